"""Constant-time variable base for secret scalars (ECGPU_SECRET_SCALARS with a point: ECDH), csrc/varbase_ct.hpp.

P-256 / P-384 run `vbct::mul_kernel` (Jacobian doublings, masked scans of per-lane affine tables, exception-free by the
fold k -> min(k, n - k)); secp256k1 runs `k256_mul_ct_kernel` (csrc/varbase_ct_k256.hpp: the reference's GLV split, Jacobian
formulas over one common-Z table per unit - exception-free by the bounds of the split -, one scan per window for both halves).  Every curve is compared with the C oracle at 2^20 units:
with 262 144 (P-384: 196 608) resident lanes that is 4 to 6 table slots per lane, so the shared table inversion and the
batched output run over several units; a second, ragged case adds a second pass.  Planted: zero, one, n - 1, the scalars a windowed
Jacobian schedule without the fold would break on (n - 2, n - 6, n - 16 .. n - 1), (n +- 1) / 2, scalars >= n, identity
points, and the same point with k and n - k.
"""
import os

import numpy as np
import pytest

from oracle import coracle as CO
from oracle import synth

pytestmark = pytest.mark.gpu
THREADS = max(1, min(os.cpu_count() or 1, 16))
ORDER = {0: synth.M.K256.n, 1: synth.M.P256.n, 2: synth.M.P384.n}


def _edge_values(n_ord, nb):
    vals = ([0, 1, 2, 8, 9, 16, (n_ord - 1) // 2, (n_ord + 1) // 2, n_ord + 3, (1 << (8 * nb)) - 1, int("8" * (2 * nb), 16) % n_ord]
            + [n_ord - d for d in range(1, 18)])
    if n_ord == synth.M.K256.n:
        # secp256k1 (csrc/varbase_ct_k256.hpp): small multiples of both GLV halves, and scalars whose split lands next to the corners of
        # the fundamental cell, where |k1|, |k2| reach their bounds - the operands closest to the lattice's shortest vector
        lam = synth.M.K256_LAMBDA
        a1, a2, b1 = 0x3086D221A7D46BCDE86C90E49284EB15, 0x114CA50F7A8E2F3F657C1108D9D44CFD8, -0xE4437ED6010E88286F547FA90ABFE4C3
        vals += [(d + e * lam) % n_ord for d in (0, 1, 8, 16, n_ord - 1, n_ord - 8) for e in (1, 8, 17, n_ord - 1)]
        for s1 in (-1, 1):
            for s2 in (-1, 1):
                x, y = (s1 * a1 + s2 * a2) // 2, (s1 * b1 + s2 * a1) // 2
                vals += [(x + d + y * lam) % n_ord for d in (-1, 0, 1)] + [(x + (y + d) * lam) % n_ord for d in (-1, 1)]
    return vals


def _run(cname, cid, n, first, sample_all):
    import torch
    import ecgpu
    ctx = ecgpu.Context(0)
    cv = ctx.curve(cname)
    nb = cv.nb
    lanes = 4 * torch.cuda.get_device_properties(0).multi_processor_count * 256
    d_s = torch.empty((n, nb), dtype=torch.uint8, device="cuda")
    d_p = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
    d_o = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
    d_i = torch.empty((n,), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_s, n, synth.SEED, first)
    cv.synth_points_device(d_p, n, synth.SEED, first)
    ctx.synchronize()
    vals = _edge_values(ORDER[cid], nb)
    planted = {}
    # spread the edge scalars over the slots of a lane (index = slot * lanes + lane) and both passes
    for j, v in enumerate(vals):
        i = (j % 8) * lanes + 1000 + 7 * j
        if i >= n:
            i = (j * 7919 + 13) % n
        planted[i] = v
    for i, v in planted.items():
        d_s[i] = torch.from_numpy(np.frombuffer(int(v).to_bytes(nb, "big"), dtype=np.uint8).copy()).cuda()
    ident = [5, lanes + 77, n - 3]
    for i in ident:
        d_p[i] = 0
    torch.cuda.synchronize()
    cv.mul_device(d_s, d_p, d_o, n, d_out_inf=d_i, flags=ecgpu.SECRET_SCALARS)
    ctx.synchronize()
    if sample_all:
        idx = np.arange(n)
    else:
        idx = np.unique(np.concatenate([np.arange(0, 4096), np.arange(n - 4096, n), np.arange(0, n, max(1, n // 32768)),
                                        np.array(sorted(planted) + ident, dtype=np.int64)]))
    t_idx = torch.from_numpy(idx).cuda()
    s = d_s[t_idx].cpu().numpy()
    p = d_p[t_idx].cpu().numpy()
    got = torch.cat([d_o[t_idx], d_i[t_idx, None]], dim=1).cpu().numpy()
    sw = s.copy()
    for i, v in planted.items():                  # the oracle takes canonical scalars
        if v >= ORDER[cid]:
            sw[np.searchsorted(idx, i)] = np.frombuffer(int(v % ORDER[cid]).to_bytes(nb, "big"), dtype=np.uint8)
    want = CO.lincomb_batch(cid, sw, p, threads=THREADS)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "units %s differ from the oracle" % idx[bad[:8]].tolist()
    for i in ident:
        r = got[np.searchsorted(idx, i)]
        assert r[-1] == 1 and not r[:-1].any(), i
    # the public-data throughput schedule must give the same bytes
    d_o2 = torch.empty_like(d_o)
    d_i2 = torch.empty_like(d_i)
    cv.mul_device(d_s, d_p, d_o2, n, d_out_inf=d_i2)
    ctx.synchronize()
    assert torch.equal(d_o, d_o2) and torch.equal(d_i, d_i2)
    ctx.close()


@pytest.mark.parametrize("cname,cid", [("p256", 1), ("p384", 2), ("k256", 0)])
def test_secret_scalar_variable_base_2p20(cname, cid):
    _run(cname, cid, 1 << 20, 70_000_000 + cid, sample_all=True)


@pytest.mark.parametrize("cname,cid", [("p256", 1), ("p384", 2)])
def test_secret_scalar_variable_base_two_passes(cname, cid):
    """2^22 + 2^18 + 777 units: a full pass of 16 slots per lane (262 144 lanes at 4 waves per SIMD on P-256, 196 608 at 3 on
    P-384) and a ragged second pass."""
    _run(cname, cid, (1 << 22) + (1 << 18) + 777, 80_000_000 + cid, sample_all=False)


@pytest.mark.parametrize("cname,cid", [("p256", 1), ("p384", 2), ("k256", 0)])
def test_ecdh_uses_the_secret_scalar_schedule(cname, cid):
    """Curve.diffie_hellman (elliptic_curve::ecdh::diffie_hellman, k256/src/ecdh.rs:41-45): both sides agree, the shared
    secret is x(k P) of the model, projective input and output work, and the reference schedule gives the same bytes."""
    import ecgpu
    c = synth.M.CURVES[cname]
    ctx = ecgpu.Context(0)
    cv = ctx.curve(cname)
    nb = cv.nb
    n = 300
    a = [synth.scalar(c, 9000 + i) or 1 for i in range(n)]
    b = [synth.scalar(c, 9500 + i) or 1 for i in range(n)]
    a[0], b[1], a[2] = 1, c.n - 1, c.n - 2
    tob = lambda v: int(v).to_bytes(nb, "big")
    pa, _ = cv.mul_by_generator(b"".join(map(tob, a)), flags=ecgpu.SECRET_SCALARS)
    pb, _ = cv.mul_by_generator(b"".join(map(tob, b)), flags=ecgpu.SECRET_SCALARS)
    s1 = cv.diffie_hellman(b"".join(map(tob, a)), pb)
    s2 = cv.diffie_hellman(b"".join(map(tob, b)), pa)
    assert bytes(s1) == bytes(s2)
    for i in (0, 1, 2, 17, n - 1):
        want = synth.M.affine_mul(c, a[i] * b[i] % c.n, (c.gx, c.gy))
        assert bytes(s1[i]) == synth.M.i2b(c, want[0]), i
    ref, _ = cv.mul(b"".join(map(tob, a)), pb, flags=ecgpu.EXACT_REFERENCE)
    assert bytes(ref[:, :nb]) == bytes(s1)
    proj = cv.mul(b"".join(map(tob, a)), pb, out_format=ecgpu.PROJECTIVE, flags=ecgpu.SECRET_SCALARS)
    if cid == 0:          # secp256k1: the reference schedule, hence the reference's own (X, Y, Z)
        xy, _ = cv.batch_normalize(proj)
        assert bytes(xy[:, :nb]) == bytes(s1)
    else:                 # (x : y : 1)
        one = (1).to_bytes(nb, "big")
        assert all(bytes(proj[i][:nb]) == bytes(s1[i]) and bytes(proj[i][2 * nb:]) == one for i in range(n))
    ctx.close()


@pytest.mark.parametrize("cname,cid", [("p256", 1), ("p384", 2), ("k256", 0)])
def test_ecdh_batch_entry_point(cname, cid):
    """ecgpu_ecdh_batch: x((public * secret).to_affine()) with the checks the reference's types perform on construction -
    NonZeroScalar (0 < d < n), PublicKey (canonical, on the curve, not the identity); host buffers, device buffers, and
    the 2^21-unit host pipeline agree."""
    import torch
    import ecgpu
    c = synth.M.CURVES[cname]
    ctx = ecgpu.Context(0)
    cv = ctx.curve(cname)
    nb = cv.nb
    n = 500
    d = CO.synth_scalars(cid, n, synth.SEED, 12_000)
    q = CO.synth_points(cid, n, synth.SEED, 12_000)
    d[3] = 0                                                                     # zero secret
    d[4] = np.frombuffer(int(c.n).to_bytes(nb, "big"), dtype=np.uint8)            # = n
    d[5] = np.frombuffer(int(c.n - 1).to_bytes(nb, "big"), dtype=np.uint8)        # fine
    q[6] = 0                                                                     # identity encoding
    q[7][-1] ^= 1                                                                # off the curve
    q[8][:nb] = np.frombuffer(int(c.p).to_bytes(nb, "big"), dtype=np.uint8)       # x = p
    shared, ok = cv.ecdh(d, q)
    bad = {3, 4, 6, 7, 8}
    assert ok.tolist() == [0 if i in bad else 1 for i in range(n)]
    want = CO.lincomb_batch(cid, d, q, threads=THREADS)
    for i in range(n):
        if i in bad:
            assert not shared[i].any(), i
        else:
            assert bytes(shared[i]) == bytes(want[i][:nb]) and want[i][-1] == 0, i
    d_d, d_q = torch.from_numpy(d).cuda(), torch.from_numpy(q).cuda()
    d_s = torch.empty((n, nb), dtype=torch.uint8, device="cuda")
    d_ok = torch.empty((n,), dtype=torch.uint8, device="cuda")
    cv.ecdh_device(d_d, d_q, d_s, d_ok, n)
    ctx.synchronize()
    assert bytes(d_s.cpu().numpy()) == bytes(shared) and bytes(d_ok.cpu().numpy()) == bytes(ok)
    if cid == 1:                                 # the chunked host pipeline (>= 2^21 units) against the device path
        m = (1 << 21) + 333
        dm = CO.synth_scalars(cid, m, synth.SEED, 13_000)
        qm = CO.synth_points(cid, m, synth.SEED, 13_000)
        dm[m - 5] = 0
        sh, okm = cv.ecdh(dm, qm)
        d_dm, d_qm = torch.from_numpy(dm).cuda(), torch.from_numpy(qm).cuda()
        d_sm = torch.empty((m, nb), dtype=torch.uint8, device="cuda")
        d_okm = torch.empty((m,), dtype=torch.uint8, device="cuda")
        cv.ecdh_device(d_dm, d_qm, d_sm, d_okm, m)
        ctx.synchronize()
        assert bytes(d_sm.cpu().numpy()) == bytes(sh) and bytes(d_okm.cpu().numpy()) == bytes(okm) and okm.sum() == m - 1
        idx = np.arange(0, m, m // 2000)
        w = CO.lincomb_batch(cid, dm[idx], qm[idx], threads=THREADS)
        assert bytes(sh[idx]) == bytes(w[:, :nb])
    ctx.close()


def _small_order_offcurve_points(c, orders=(5, 17)):
    """points of small order on the singular cubic y^2 = x^3 - 3x - 2 (see tests/test_hosttwin_vbct.py)"""
    import random
    M = synth.M
    p = c.p
    N = p - 1 if pow((-3) % p, (p - 1) // 2, p) == 1 else p + 1
    rng = random.Random(20260405)
    out = []
    for q in orders:
        while True:
            x = rng.randrange(p)
            t = (x - 2) % p
            if pow(t, (p - 1) // 2, p) != 1:
                continue
            P = M.affine_mul(c, N // q, (x, (x + 1) * M.field_sqrt(c, t) % p))
            if P is not None:
                out.append(P)
                break
    return out


@pytest.mark.parametrize("cname,cid", [("p256", 1), ("p384", 2), ("k256", 0)])
def test_invalid_public_key_does_not_poison_its_neighbours(cname, cid):
    """a public key that is not on the curve - here points of order 5 and 17 on the singular cubic
    y^2 = x^3 - 3x - 2, which walk the raw mixed addition into acc = +-Q so that Z = 0 - used to zero the VALID results that
    shared its lane's output inversion, while those still reported ok = 1.  2^20 + 999 units: every lane holds several units.
    (a) ecgpu_ecdh_batch: the bad keys get ok = 0 and zeros, every other unit is the oracle's; (b) ecgpu_mul_batch with
    ECGPU_SECRET_SCALARS, which does not validate: the bad units' output is unspecified, every other unit is untouched."""
    import torch
    import ecgpu
    c = synth.M.CURVES[cname]
    ctx = ecgpu.Context(0)
    cv = ctx.curve(cname)
    nb = cv.nb
    n = (1 << 20) + 999
    d = CO.synth_scalars(cid, n, synth.SEED, 21_000)
    q = CO.synth_points(cid, n, synth.SEED, 21_000)
    bad_pts = _small_order_offcurve_points(c) if cid else [(4, 8), (9, 27)]       # secp256k1: points of y^2 = x^3
    lanes = 4 * torch.cuda.get_device_properties(0).multi_processor_count * 256
    bad_idx = [7, lanes + 7, 2 * lanes + 70001, n - 2, 123456, 123457]
    for j, i in enumerate(bad_idx):
        P = bad_pts[j % len(bad_pts)]
        q[i] = np.frombuffer(synth.M.i2b(c, P[0]) + synth.M.i2b(c, P[1]), dtype=np.uint8)
    d_d, d_q = torch.from_numpy(d).cuda(), torch.from_numpy(q).cuda()
    d_s = torch.empty((n, nb), dtype=torch.uint8, device="cuda")
    d_ok = torch.empty((n,), dtype=torch.uint8, device="cuda")
    cv.ecdh_device(d_d, d_q, d_s, d_ok, n)
    d_o = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
    d_i = torch.empty((n,), dtype=torch.uint8, device="cuda")
    cv.mul_device(d_d, d_q, d_o, n, d_out_inf=d_i, flags=ecgpu.SECRET_SCALARS)
    # the public-data schedule on the valid units only is the reference for ALL of them (itself checked against the oracle below)
    q_ok = q.copy()
    for i in bad_idx:
        q_ok[i] = q[0]
    d_o2 = torch.empty_like(d_o)
    cv.mul_device(d_d, torch.from_numpy(q_ok).cuda(), d_o2, n)
    ctx.synchronize()
    shared, ok, prod, ref = d_s.cpu().numpy(), d_ok.cpu().numpy(), d_o.cpu().numpy(), d_o2.cpu().numpy()
    keep = np.ones(n, dtype=bool)
    keep[bad_idx] = False
    assert not ok[bad_idx].any() and ok[keep].all() and not shared[bad_idx].any()
    assert (shared[keep] == ref[keep][:, :nb]).all(), "a valid unit's shared secret changed"
    assert (prod[keep] == ref[keep]).all() and not d_i.cpu().numpy()[keep].any(), "a valid unit's product changed"
    near = np.unique(np.clip(np.concatenate([np.array(bad_idx) + k * lanes for k in range(-3, 4)] + [np.arange(0, n, n // 512)]), 0, n - 1))
    near = near[keep[near]]
    want = CO.lincomb_batch(cid, d[near], q[near], threads=THREADS)
    assert bytes(ref[near]) == bytes(want[:, :2 * nb])
    ctx.close()


@pytest.mark.parametrize("cname,cid", [("p256", 1), ("p384", 2), ("k256", 0)])
def test_no_secret_stays_in_the_workspaces(cname, cid):
    """include/ecgpu.h promises that the intermediate products of ecgpu_ecdh_batch do not stay in the context's
    workspace.  The constant-time kernels park each result's Jacobian (X, Y, Z) and the prefix products of the output inversion in the
    table workspace: they clear them themselves now.  After a call, neither the table workspace nor the ECDH workspace nor the
    staging slots may hold a shared x, a secret scalar, or anything but public table data: checked by searching the workspaces for
    the 32 / 48-byte strings (wire form and limb-reversed internal form) and by the parked-result regions being all zero."""
    import torch
    import ecgpu
    ctx = ecgpu.Context(0)
    cv = ctx.curve(cname)
    nb = cv.nb
    n = 70_000
    d = CO.synth_scalars(cid, n, synth.SEED, 31_000)
    q = CO.synth_points(cid, n, synth.SEED, 31_000)
    shared, ok = cv.ecdh(d, q)                    # host buffers: staging slots 0 (secrets) and 2 (shared values) are used
    assert ok.all()

    dump = ctx.debug_workspace

    tab, ws, st0, st2 = dump(0), dump(1), dump(16 + 0), dump(16 + 2)
    assert len(tab) > 0 and len(ws) >= n * 4 * nb
    assert not any(st0) and not any(st2), "staged secrets / shared values were left behind"
    assert not any(ws[:n * 2 * nb]), "the products (shared points) were left in the ECDH workspace"
    words = lambda b: b"".join(b[i:i + 4][::-1] for i in range(len(b) - 4, -1, -4))      # big-endian bytes -> little-endian 32-bit limbs, least significant first
    for i in (0, 1, 63, 64, 255, 256, 4097, n - 1):
        sx, dk = bytes(shared[i]), bytes(d[i])
        for name, blob in (("table workspace", tab), ("ECDH workspace", ws)):
            assert sx not in blob and words(sx) not in blob, (name, i, "shared x")
            assert dk not in blob and words(dk) not in blob, (name, i, "secret scalar")
    ctx.close()
