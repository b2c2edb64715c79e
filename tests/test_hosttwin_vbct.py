"""Host build of the constant-time variable-base body (csrc/varbase_ct.hpp: the kernel behind ECGPU_SECRET_SCALARS with a
variable base, i.e. ECDH on P-256 / P-384), walked with a handful of lanes over a lane-interleaved workspace:

* (P-256 / P-384: Jacobian formulas made exception-free by the fold; secp256k1: Jacobian formulas, exception-free by the bounds of the GLV split)
* k P against the big-integer model for random and edge scalars - including every scalar for which a windowed
  Jacobian schedule WITHOUT the fold k -> min(k, n - k) would meet P + P or P - P in its last addition (n - 2 for
  P-256, n - 6 for P-384, and the whole range n - 16 .. n - 1), zero, identity inputs, scalars >= n;
* the sequence of table entries the window loop reads is the same for every scalar (all eight entries of the unit's
  table, 8 NW + 1 times), which is the host-side half of the constant-time evidence (the device half is
  profiles/r03_ct_counters.txt: identical instruction counters for different scalar sets).
"""
import ctypes
import random

import pytest

from oracle import ecmodel as M
from oracle import synth
from hosttwin_util import lib, buf, outbuf

CURVES = [("p256", 1), ("p384", 2), ("k256", 0)]     # k256: csrc/varbase_ct_k256.hpp (GLV + Jacobian, lattice argument), the others csrc/varbase_ct.hpp


def _vbct(cid, c, ks, ps, lanes, out_fmt=0, proj_in=False, fn="ht_vbct_mul"):
    nb = c.nbytes
    n = len(ks)
    sb = b"".join(int(k).to_bytes(nb, "big") for k in ks)
    if proj_in:
        pb = b"".join(M.proj_bytes(c, p) for p in ps)
    else:
        pb = b"".join(M.i2b(c, p[0]) + M.i2b(c, p[1]) if p is not None else bytes(2 * nb) for p in ps)
    out = outbuf((3 if out_fmt else 2) * nb * n)
    inf = outbuf(n)
    L = lib()
    f = getattr(L, fn)
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]
    assert f(cid, buf(sb), buf(pb), 1 if proj_in else 0, out, out_fmt, inf, n, lanes) == 0
    return bytes(out), bytes(inf)


K256_LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72      # k256/src/arithmetic/mul.rs:129-152


def edge_scalars(c):
    n = c.n
    return ([0, 1, 2, 7, 8, 9, 15, 16, 17, 0x88, 0x87, 0x89, (n - 1) // 2, (n + 1) // 2, (n + 3) // 2, (n - 3) // 2, n, n + 5,
             1 << (8 * c.nbytes - 1), (1 << (8 * c.nbytes)) - 1, int("8" * (2 * c.nbytes), 16) % n, int("7" * (2 * c.nbytes), 16) % n]
            + [n - d for d in range(1, 18)]
            # around the GLV split of secp256k1 (harmless extra cases for the others): 2^128 +- 1, lambda (k1 = 0, k2 = 1), lambda +- 1, -lambda
            + [(1 << 128) - 1, 1 << 128, (1 << 128) + 1, K256_LAMBDA % n, (K256_LAMBDA + 1) % n, (K256_LAMBDA - 1) % n, (n - K256_LAMBDA) % n]
            # small multiples of both halves and their sums (prefixes (d, 0), (0, e), (d, e): where an incomplete addition would meet +-Q first)
            + [(d + e * K256_LAMBDA) % n for d in (0, 1, 2, 8, 15, 16, 17, n - 1, n - 8, n - 16) for e in (1, 2, 8, 16, 17, n - 1, n - 8)])


def k256_corner_scalars():
    """Scalars whose GLV split lands next to the corners of the fundamental cell, where |k1| and |k2| reach their bounds
    (tests/test_oracle_golden.py::test_k256_glv_bounds): the operands closest to the lattice's shortest vector."""
    n, lam = M.K256.n, K256_LAMBDA
    a1, a2, b1 = 0x3086D221A7D46BCDE86C90E49284EB15, 0x114CA50F7A8E2F3F657C1108D9D44CFD8, -0xE4437ED6010E88286F547FA90ABFE4C3
    out = []
    for s1 in (-1, 1):
        for s2 in (-1, 1):
            x, y = (s1 * a1 + s2 * a2) // 2, (s1 * b1 + s2 * a1) // 2
            out += [(x + d + y * lam) % n for d in (-9, -1, 0, 1, 2, 8)] + [(x + (y + d) * lam) % n for d in (-8, -1, 1, 7)]
    return out


def test_vbct_k256_corner_scalars():
    c = M.CURVES["k256"]
    nb = c.nbytes
    ks = k256_corner_scalars()
    ps = [synth.point(c, 4300 + (i % 3)) for i in range(len(ks))]
    out, inf = _vbct(0, c, ks, ps, 3, fn="ht_vbct_mul16")
    for i, k in enumerate(ks):
        want = M.affine_mul(c, k % c.n, ps[i])
        assert out[2 * nb * i:2 * nb * (i + 1)] == M.i2b(c, want[0]) + M.i2b(c, want[1]) and inf[i] == 0, (i, hex(k))


@pytest.mark.parametrize("cn,cid", CURVES)
@pytest.mark.parametrize("lanes", [1, 3])
def test_vbct_edge_and_random_scalars(cn, cid, lanes):
    c = M.CURVES[cn]
    nb = c.nbytes
    ks = edge_scalars(c) + [synth.scalar(c, 4000 + i) for i in range(11)]
    n = len(ks)
    ps = [synth.point(c, 4100 + (i % 5)) for i in range(n)]
    ps[3] = None                                  # identity inputs, one of them with an edge scalar
    ps[n - 2] = None
    out, inf = _vbct(cid, c, ks, ps, lanes)
    for i in range(n):
        want = None if ps[i] is None else M.affine_mul(c, ks[i] % c.n, ps[i])
        got = out[2 * nb * i:2 * nb * (i + 1)]
        if want is None:
            assert got == bytes(2 * nb) and inf[i] == 1, (i, hex(ks[i]))
        else:
            assert got == M.i2b(c, want[0]) + M.i2b(c, want[1]) and inf[i] == 0, (i, hex(ks[i]))
    # projective in / projective out: (x : y : 1), identity (0 : 1 : 0)
    rng = random.Random(77 + cid)
    pp = [M.IDENTITY if p is None else (lambda z: (p[0] * z % c.p, p[1] * z % c.p, z))(rng.randrange(1, c.p)) for p in ps]
    outp, _ = _vbct(cid, c, ks, pp, lanes, out_fmt=1, proj_in=True)
    for i in range(n):
        want = None if ps[i] is None else M.affine_mul(c, ks[i] % c.n, ps[i])
        got = outp[3 * nb * i:3 * nb * (i + 1)]
        if want is None:
            assert got == M.proj_bytes(c, M.IDENTITY), i
        else:
            assert got == M.i2b(c, want[0]) + M.i2b(c, want[1]) + M.i2b(c, 1), i


@pytest.mark.parametrize("cn,cid", CURVES)
@pytest.mark.parametrize("lanes,n", [(3, 53), (2, 1), (5, 40)])
def test_vbct_slot_counts_and_passes(cn, cid, lanes, n):
    """(3, 53): two full passes and a ragged third; (2, 1): a single unit; (5, 40): exactly one pass."""
    c = M.CURVES[cn]
    nb = c.nbytes
    ks = [synth.scalar(c, 5000 + i) for i in range(n)]
    ps = [synth.point(c, 5000 + i) for i in range(n)]
    if n > 30:
        ks[lanes * 2] = 0
        ks[min(lanes * 9, n - 1)] = c.n - 2
        ps[lanes * 4 + 1] = None
    out, inf = _vbct(cid, c, ks, ps, lanes)
    for i in range(n):
        want = None if ps[i] is None else M.affine_mul(c, ks[i] % c.n, ps[i])
        got = out[2 * nb * i:2 * nb * (i + 1)]
        if want is None:
            assert got == bytes(2 * nb) and inf[i] == 1, i
        else:
            assert got == M.i2b(c, want[0]) + M.i2b(c, want[1]) and inf[i] == 0, i


@pytest.mark.parametrize("cn,cid", CURVES)
@pytest.mark.parametrize("fn", ["ht_vbct_mul16", "ht_vb_mul16", "ht_vb_mul_w5"])
def test_sixteen_slots_per_pass(cn, cid, fn):
    if cid == 0 and fn != "ht_vbct_mul16":
        pytest.skip("the public-data k256 kernel is mulfast_k256.hpp (tests/test_hosttwin_k256_fast.py)")
    """The product's pass size (16 table slots per lane and pass): 2 lanes, 53 units = one full pass and a ragged second one
    (11 and 10 slots), for the constant-time body and for the public-data body."""
    c = M.CURVES[cn]
    nb = c.nbytes
    n, lanes = 53, 2
    ks = [synth.scalar(c, 7000 + i) for i in range(n)]
    ps = [synth.point(c, 7000 + i) for i in range(n)]
    ks[31], ks[33], ks[40], ps[17], ps[50] = 0, c.n - 2, c.n - 6, None, None
    out, inf = _vbct(cid, c, ks, ps, lanes, fn=fn)
    for i in range(n):
        want = None if ps[i] is None else M.affine_mul(c, ks[i] % c.n, ps[i])
        got = out[2 * nb * i:2 * nb * (i + 1)]
        if want is None:
            assert got == bytes(2 * nb) and inf[i] == 1, i
        else:
            assert got == M.i2b(c, want[0]) + M.i2b(c, want[1]) and inf[i] == 0, i


def _trace(fn):
    L = lib()
    L.ht_trace_stop.restype = ctypes.c_size_t
    L.ht_trace_stop.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    L.ht_trace_start()
    fn()
    cap = 1 << 16
    arr = (ctypes.c_int * cap)()
    cnt = L.ht_trace_stop(arr, cap)
    assert cnt <= cap
    return list(arr[:cnt])


@pytest.mark.parametrize("cn,cid", CURVES)
def test_vbct_reads_the_whole_table_for_every_scalar(cn, cid):
    c = M.CURVES[cn]
    nb = c.nbytes
    P = synth.point(c, 6000)
    traces = []
    for k in [0, 1, c.n - 1, c.n - 2, c.n - 6, (c.n - 1) // 2, int("f" * (2 * nb), 16) % c.n, int("8" * (2 * nb), 16) % c.n,
              synth.scalar(c, 6001), synth.scalar(c, 6002)]:
        traces.append(_trace(lambda: _vbct(cid, c, [k], [P], 1)))
    nwin = 33 if cid == 0 else 2 * nb + 1               # k256: 32 nibbles of a GLV half and the carry digits (one scan serves both halves); else 8 NW nibbles and the carry digit
    assert all(t == traces[0] for t in traces)
    assert traces[0] == list(range(8)) * nwin


def small_order_offcurve_points(c, orders=(5, 17)):
    """Points of small order on the singular cubic y^2 = x^3 - 3x - 2 = (x + 1)^2 (x - 2): not on the curve, yet the a = -3
    formulas (which never use b) act on them as a group of order p - 1 or p + 1.  A windowed schedule walks such a point into
    acc = +-Q within a few digits, so the raw mixed addition leaves Z = 0 (one such key in an ECDH batch used
    to zero the valid neighbours that shared its lane's output inversion)."""
    p = c.p
    N = p - 1 if pow((-3) % p, (p - 1) // 2, p) == 1 else p + 1
    rng = random.Random(20260405)
    out = []
    for q in orders:
        assert N % q == 0
        while True:
            x = rng.randrange(p)
            t = (x - 2) % p
            if pow(t, (p - 1) // 2, p) != 1:
                continue
            P = M.affine_mul(c, N // q, (x, (x + 1) * M.field_sqrt(c, t) % p))
            if P is not None:
                assert M.affine_mul(c, q, P) is None and not M.on_curve(c, P)
                out.append(P)
                break
    return out


@pytest.mark.parametrize("cn,cid", CURVES)
@pytest.mark.parametrize("fn", ["ht_vbct_mul16", "ht_vbct_mul"])
def test_offcurve_small_order_point_does_not_poison_its_lane(cn, cid, fn):
    """One lane, so every unit of the pass shares ONE output inversion with the invalid points: each valid neighbour must come back
    bit-exact.  (secp256k1 has a = 0: its singular companion is y^2 = x^3; the same planted inputs exercise its Z guard.)"""
    c = M.CURVES[cn]
    nb = c.nbytes
    if cid == 0:
        bad = [(4, 8), (9, 27)]                    # on y^2 = x^3 (additive group: every point has order p); plus x = 0 cases below
    else:
        bad = small_order_offcurve_points(c)
    n = 8
    ks = [synth.scalar(c, 9100 + i) for i in range(n)]
    ps = [synth.point(c, 9100 + i) for i in range(n)]
    ps[2], ps[5] = bad[0], bad[1]
    out, inf = _vbct(cid, c, ks, ps, 1, fn=fn)
    for i in range(n):
        if i in (2, 5):
            continue                               # unspecified output for input that violates the reference's type invariant
        want = M.affine_mul(c, ks[i] % c.n, ps[i])
        assert out[2 * nb * i:2 * nb * (i + 1)] == M.i2b(c, want[0]) + M.i2b(c, want[1]) and inf[i] == 0, i


@pytest.mark.parametrize("cn,cid", [("p256", 1), ("p384", 2)])
def test_five_bit_windows_edge_scalars(cn, cid):
    """The 5-bit-window form of the public-data variable-base body (csrc/varbase_lane.hpp, WB = 5: 16 table entries, digits = 5-bit
    fields of k' + 0x..108421 minus 16): every edge scalar of the recoding - digits -16 and 15, fields that straddle a word boundary,
    the top field, k' = (n - 1) / 2 - and random ones, against the model; identity inputs; two lanes, three passes."""
    c = M.CURVES[cn]
    nb = c.nbytes
    ks = edge_scalars(c) + [int("f" * (2 * nb), 16) % c.n, int("0" + "f" * (2 * nb - 1), 16) % c.n] + [16 * 32 ** j for j in (0, 1, 6, 7, 12, 13, 50)] + \
         [(31 * 32 ** j + 15) % c.n for j in (1, 6, 12, 51)] + [synth.scalar(c, 8000 + i) for i in range(20)]
    n = len(ks)
    ps = [synth.point(c, 8100 + (i % 7)) for i in range(n)]
    ps[4] = None
    out, inf = _vbct(cid, c, ks, ps, 2, fn="ht_vb_mul_w5")
    for i in range(n):
        want = None if ps[i] is None else M.affine_mul(c, ks[i] % c.n, ps[i])
        got = out[2 * nb * i:2 * nb * (i + 1)]
        if want is None:
            assert got == bytes(2 * nb) and inf[i] == 1, (i, hex(ks[i]))
        else:
            assert got == M.i2b(c, want[0]) + M.i2b(c, want[1]) and inf[i] == 0, (i, hex(ks[i]))
