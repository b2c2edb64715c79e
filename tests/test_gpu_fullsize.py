"""BASELINE.json's full sizes through the code paths they really take (first and last 4 096
units, a strided sample of >= 32 768, all against the C oracle on the same seeded inputs).

The variable-base kernels launch at most 4 workgroups per CU (262 144 lanes on 256 CUs); below that many units every
lane holds ONE result.  Only at the BASELINE sizes does a lane carry several results per pass (slots b > 0 of its
table workspace, the shared table inversion over all of them, the batched output inversion) and come back for a
second pass:
    config 5  p384  2^22 units:  16 slots per lane, exactly one pass; 2^22 + 2^20 + 321: a ragged second pass
    p256      2^22 + 2^19 units: 16 slots, then a ragged second pass of 2
    config 2  k256  2^24 units:  32 results per lane per pass, two passes
Edge inputs (zero scalar, identity point, scalar >= n) are planted in slots b > 0 and in the second pass.
"""
import os

import numpy as np
import pytest

from oracle import coracle as CO
from oracle import synth

pytestmark = pytest.mark.gpu

THREADS = max(1, min(os.cpu_count() or 1, 16))


def _sample_indices(n, lanes, per_pass):
    head = np.arange(0, 4096)
    tail = np.arange(n - 4096, n)
    stride = np.arange(0, n, max(1, n // 32768))
    # the first units of every slot of both passes (slot b of pass q starts at q * per_pass + b * lanes)
    slots = np.concatenate([np.arange(s, min(s + 8, n)) for s in range(0, n, lanes)])
    return np.unique(np.concatenate([head, tail, stride, slots]))


def _run_varbase(cname, cid, n, first, edges):
    import torch
    import ecgpu
    ctx = ecgpu.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    cv = ctx.curve(cname)
    nb = cv.nb
    lanes = 4 * torch.cuda.get_device_properties(0).multi_processor_count * 256
    d_s = torch.empty((n, nb), dtype=torch.uint8, device="cuda")
    d_p = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
    d_o = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
    d_i = torch.empty((n,), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_s, n, synth.SEED, first)
    cv.synth_points_device(d_p, n, synth.SEED, first)
    order = {0: synth.M.K256.n, 1: synth.M.P256.n, 2: synth.M.P384.n}[cid]
    planted = []
    for slot, pass_, lane, kind in edges(lanes):
        i = pass_ * lanes * (32 if cid == 0 else 16) + slot * lanes + lane
        if i >= n:
            continue
        if kind == "zero":
            d_s[i] = 0
        elif kind == "ident":
            d_p[i] = 0
        elif kind == "n-1":
            d_s[i] = torch.from_numpy(np.frombuffer((order - 1).to_bytes(nb, "big"), dtype=np.uint8).copy()).cuda()
        elif kind == "n+3":                      # reduced once, like Reduce<U256>::reduce
            d_s[i] = torch.from_numpy(np.frombuffer((order + 3).to_bytes(nb, "big"), dtype=np.uint8).copy()).cuda()
        planted.append((i, kind))
    torch.cuda.synchronize()
    cv.mul_device(d_s, d_p, d_o, n, d_out_inf=d_i)
    ctx.synchronize()
    idx = np.unique(np.concatenate([_sample_indices(n, lanes, lanes * (32 if cid == 0 else 16)), np.array([i for i, _ in planted], dtype=np.int64)]))
    t_idx = torch.from_numpy(idx).cuda()
    s = d_s[t_idx].cpu().numpy()
    p = d_p[t_idx].cpu().numpy()
    got = torch.cat([d_o[t_idx], d_i[t_idx, None]], dim=1).cpu().numpy()
    sw = s.copy()
    # the oracle takes canonical scalars: reduce the planted n + 3 by hand
    for i, kind in planted:
        if kind == "n+3":
            sw[np.searchsorted(idx, i)] = np.frombuffer((3).to_bytes(nb, "big"), dtype=np.uint8)
    want = CO.lincomb_batch(cid, sw, p, threads=THREADS)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "units %s differ from the oracle" % idx[bad[:8]].tolist()
    pos = {i: np.searchsorted(idx, i) for i, _ in planted}
    for i, kind in planted:
        if kind in ("zero", "ident"):
            assert got[pos[i], -1] == 1 and not got[pos[i], :-1].any(), (i, kind)
        else:
            assert got[pos[i], -1] == 0, (i, kind)
    assert int(d_i.sum().item()) == sum(1 for _, k in planted if k in ("zero", "ident"))
    ctx.close()
    return len(idx)


def _edges(lanes):
    # (slot, pass, lane, kind)
    return [(3, 0, 5, "zero"), (5, 0, 77, "ident"), (7, 0, lanes - 1, "n-1"), (1, 0, 64, "n+3"),
            (6, 1, 77, "zero"), (2, 1, 12345, "ident"), (1, 1, 1, "n-1"), (0, 1, 0, "n+3")]


def test_p384_config5_2p22():
    """BASELINE config 5 at its own size: vb::mul_kernel<CurveP384, 16, 4>, 16 slots per lane, one full pass (the planted
    "second pass" indices fall into slots 8..15)."""
    assert _run_varbase("p384", 2, 1 << 22, 40_000_000, _edges) >= 32768 + 8192


def test_p384_varbase_second_pass():
    """p384 with a full pass of 16 slots and a ragged second pass (4 slots and a partial fifth)."""
    assert _run_varbase("p384", 2, (1 << 22) + (1 << 20) + 321, 41_000_000, _edges) >= 32768 + 8192


def test_p256_varbase_2p21_plus():
    """p256 variable base with a full pass of 16 slots and a ragged second pass (2 slots and a partial third)."""
    assert _run_varbase("p256", 1, (1 << 22) + (1 << 19) + 12345, 50_000_000, _edges) >= 32768 + 8192


def test_k256_config2_2p24_second_pass():
    """BASELINE config 2 at its own size: k256_mul_fast_kernel<32, 4>, 32 results per lane and pass, two passes
    (kernels.hpp: base += T * BATCH)."""
    def edges(lanes):
        return [(3, 0, 5, "zero"), (17, 0, 77, "ident"), (31, 0, lanes - 1, "n-1"), (9, 0, 64, "n+3"),
                (30, 1, 77, "zero"), (2, 1, 12345, "ident"), (31, 1, lanes - 1, "n-1"), (0, 1, 0, "n+3")]
    assert _run_varbase("k256", 0, 1 << 24, 60_000_000, edges) >= 32768 + 8192


@pytest.mark.parametrize("cname,cid,n,first", [
    ("k256", 0, 1 << 23, 5),                       # BASELINE config 4, one GPU's share: 19-bit windows, one slab
    ("k256", 0, (1 << 24) + 777, 1 << 30),         # 16-bit windows, two slabs (24-bit term index), the second one tiny
    ("p256", 1, 1 << 22, 9),                       # 19-bit windows without the endomorphism (14 windows)
    ("k256", 0, 1 << 26, 1 << 33),                 # BASELINE config 4's TOTAL size on one card: four slabs of the 16-bit path
])
def test_msm_full_sizes_structured(cname, cid, n, first):
    """The MSM at sizes no term-by-term oracle reaches: P_i = (a0 + i d) G (computed on the device by the fixed-base path),
    seeded scalars, and the closed form (sum k_i (a0 + i d) mod n) G of SURVEY.md section 8d - over ALL terms, with a
    zero scalar, a scalar n - 1 and an identity point planted (the identity's term drops out of the closed form)."""
    import torch
    import ecgpu
    import bench
    c = {0: synth.M.K256, 1: synth.M.P256}[cid]
    ctx = ecgpu.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    cv = ctx.curve(cname)
    ps = bench.structured_point_scalars(first, n)
    d_pts = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    d_ps = torch.from_numpy(ps).cuda()
    del ps
    cv.mul_device(d_ps, None, d_pts, n)
    ctx.synchronize()
    del d_ps
    d_s = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_s, n, synth.SEED, first)
    ctx.synchronize()
    d_s[12345] = 0
    d_s[n - 7] = torch.from_numpy(np.frombuffer((c.n - 1).to_bytes(32, "big"), dtype=np.uint8).copy()).cuda()
    d_pts[n // 2 + 1] = 0
    torch.cuda.synchronize()
    d_out = torch.empty((64,), dtype=torch.uint8, device="cuda")
    cv.msm_device(d_s, d_pts, n, d_out)
    ctx.synchronize()
    ks = d_s.cpu().numpy()
    ks[n // 2 + 1] = 0                             # the identity point contributes nothing
    tot = bench.msm_expected_scalar(ks, first, c.n)
    want = synth.M.affine_mul(c, tot, (c.gx, c.gy))
    assert bytes(d_out.cpu().numpy()) == synth.M.i2b(c, want[0]) + synth.M.i2b(c, want[1])
    ctx.close()


@pytest.mark.parametrize("cname,cid,n", [("k256", 0, (1 << 23) + 4097), ("p256", 1, (1 << 24) + 5), ("p384", 2, (1 << 23) + 4097)])
def test_fixed_base_wide_tables(cname, cid, n):
    """mul_by_generator at the size classes that take the 24-bit-window generator table (2^23 + 4 097 results: 5.9 GB;
    13.7 GB and a carry window for P-384) and the 26-bit one (2^24 + 5 results: 21.5 GB), against the C oracle on the head,
    the tail and a strided sample, with a zero scalar, n - 1, a scalar >= n and a scalar just above n / 2 (the sign fold)
    planted; and against the narrower tables on prefixes (the same scalars through the smaller size classes must give the
    same bytes)."""
    import torch
    import ecgpu
    c = {0: synth.M.K256, 1: synth.M.P256, 2: synth.M.P384}[cid]
    ctx = ecgpu.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    cv = ctx.curve(cname)
    nb = cv.nb
    d_s = torch.empty((n, nb), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_s, n, synth.SEED, 77)
    ctx.synchronize()

    def row(v):
        return torch.from_numpy(np.frombuffer(int(v).to_bytes(nb, "big"), dtype=np.uint8).copy()).cuda()
    d_s[5] = 0
    d_s[6] = row(c.n - 1)
    d_s[7] = row(c.n + 3)
    d_s[n - 3] = row(c.n // 2 + 1)
    d_s[n - 2] = row(c.n // 2)
    d_s[n - 1] = row((1 << 24) - 1)                 # one full window of ones: a carry into the next window
    d_s[n - 4] = row((1 << 52) - (1 << 25))         # the same for the 26-bit windows
    torch.cuda.synchronize()
    d_o = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
    d_i = torch.empty((n,), dtype=torch.uint8, device="cuda")
    cv.mul_device(d_s, None, d_o, n, d_out_inf=d_i)
    ctx.synchronize()
    idx = np.unique(np.concatenate([np.arange(0, 2048), np.arange(n - 2048, n), np.arange(0, n, n // 8192)]))
    s = d_s.cpu().numpy()
    want = CO.lincomb_batch(cid, np.ascontiguousarray(s[idx]), None, threads=THREADS)
    got = np.concatenate([d_o.cpu().numpy()[idx], d_i.cpu().numpy()[idx, None]], axis=1)
    assert bytes(got) == bytes(want)
    for m in ((1 << 23, 1 << 21, 1 << 18) if n > (1 << 24) else (1 << 21, 1 << 18)):      # the narrower tables on prefixes
        d_o2 = torch.empty((m, 2 * nb), dtype=torch.uint8, device="cuda")
        cv.mul_device(d_s[:m], None, d_o2, m)
        ctx.synchronize()
        assert torch.equal(d_o2, d_o[:m]), m
    ctx.close()
