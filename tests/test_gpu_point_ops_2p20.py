"""Exact (X, Y, Z) of ProjectivePoint::{add, add_mixed, double} on 2^20 random pairs per curve (SURVEY.md section 8d:
"projective (X, Y, Z) exactness of add / double checked separately on 2^20 random pairs"), GPU against the C oracle's
limb-for-limb restatement of k256/src/arithmetic/projective.rs:96-274 and primeorder/src/point_arithmetic.rs:209-317.

Operands are projective points with Z != 1 (the exact outputs of reference-schedule multiplications), so every formula
runs on full-width coordinates; planted: the identity on either side, P + P, P + (-P), and identity affine operands."""
import os

import numpy as np
import pytest

from oracle import coracle as CO
from oracle import synth

pytestmark = pytest.mark.gpu
N = 1 << 20


@pytest.mark.parametrize("cname,cid", [("k256", 0), ("p256", 1), ("p384", 2)])
def test_exact_xyz_on_2p20_pairs(cname, cid):
    import torch
    import ecgpu
    ctx = ecgpu.Context(0)
    cv = ctx.curve(cname)
    nb = cv.nb
    u8 = dict(dtype=torch.uint8, device="cuda")
    d_s = torch.empty((2 * N, nb), **u8)
    d_b = torch.empty((2 * N, 2 * nb), **u8)
    d_pq = torch.empty((2 * N, 3 * nb), **u8)
    cv.synth_scalars_device(d_s, 2 * N, synth.SEED, 31_000_000)
    cv.synth_points_device(d_b, 2 * N, synth.SEED, 31_000_000)
    # random projective representatives: the reference schedule's own (X, Y, Z) of k * B
    cv.mul_device(d_s, d_b, d_pq, 2 * N, out_format=ecgpu.PROJECTIVE, flags=ecgpu.EXACT_REFERENCE)
    d_p, d_q = d_pq[:N], d_pq[N:]
    d_qa = d_b[N:].clone()                          # affine operands of add_mixed
    ident = torch.zeros((3 * nb,), **u8)
    ident[2 * nb - 1] = 1                           # (0 : 1 : 0)
    d_p[11] = ident
    d_q[12] = ident
    d_p[13] = ident; d_q[13] = ident
    d_q[14] = d_p[14]                               # P + P through the addition formula
    cv.mul_device(d_s[N + 15:N + 16], d_b[N + 15:N + 16], d_q[15:16], 1, out_format=ecgpu.PROJECTIVE, flags=ecgpu.EXACT_REFERENCE)
    d_p[15] = d_q[15]
    d_q[15, nb:2 * nb] = torch.from_numpy(np.frombuffer(
        ((synth.M.CURVES[cname].p - int.from_bytes(bytes(d_q[15, nb:2 * nb].cpu().numpy()), "big")) % synth.M.CURVES[cname].p).to_bytes(nb, "big"),
        dtype=np.uint8).copy()).cuda()              # Q = -P
    d_qa[16] = 0                                    # AffinePoint::IDENTITY
    d_qa[17] = 0; d_p[17] = ident
    d_o = {k: torch.empty((N, 3 * nb), **u8) for k in ("add", "dbl", "mixed")}
    cv.add_device(d_p, d_q, d_o["add"], N)
    ctx.check(ctx.lib.ecgpu_point_double_batch(ctx.handle, cid, d_p.data_ptr(), d_o["dbl"].data_ptr(), N, ecgpu.DEVICE))
    ctx.check(ctx.lib.ecgpu_point_add_mixed_batch(ctx.handle, cid, d_p.data_ptr(), d_qa.data_ptr(), d_o["mixed"].data_ptr(), N, ecgpu.DEVICE))
    ctx.synchronize()
    p, q, qa = d_p.cpu().numpy(), d_q.cpu().numpy(), d_qa.cpu().numpy()
    got = {k: v.cpu().numpy() for k, v in d_o.items()}
    # the C oracle, one slice per host core
    from concurrent.futures import ThreadPoolExecutor
    T = max(1, min(os.cpu_count() or 1, 16))
    cuts = [N * t // T for t in range(T + 1)]

    def work(t):
        lo, hi = cuts[t], cuts[t + 1]
        return (CO.point_op(cid, 0, p[lo:hi], q[lo:hi]), CO.point_op(cid, 1, p[lo:hi]), CO.point_op(cid, 2, p[lo:hi], qa[lo:hi]))

    CO.lib()
    with ThreadPoolExecutor(T) as ex:              # ctypes drops the GIL in the C calls
        parts = list(ex.map(work, range(T)))
    for j, key in enumerate(("add", "dbl", "mixed")):
        want = np.concatenate([pt[j] for pt in parts])
        bad = np.nonzero((got[key] != want).any(axis=1))[0]
        assert bad.size == 0, (key, bad[:8].tolist())
    assert bytes(got["add"][13]) == bytes(ident.cpu().numpy()) and bytes(got["mixed"][17]) == bytes(ident.cpu().numpy())
    assert not got["add"][15][2 * nb:].any()        # P + (-P): Z = 0
    ctx.close()
