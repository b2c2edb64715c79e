"""Randomised cross-check of the three schedules of `scalar * point` on the device: secret-scalar kernels (ECGPU_SECRET_SCALARS:
vbct::mul_kernel / k256_mul_ct_kernel), the reference schedule (ECGPU_EXACT_REFERENCE) and the public-data throughput schedule must
give the same bytes for every batch size (ragged passes, sizes around the lane count) and input mix (edge scalars, identity points,
repeated points, projective input); ecgpu_ecdh_batch must agree on the valid units.   python tools/ct_stress.py [cases] [seed]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import numpy as np
import torch
import ecgpu
from oracle import ecmodel as M, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
ctx = ecgpu.Context(0)
lanes = 4 * torch.cuda.get_device_properties(0).multi_processor_count * 256
bad = 0
for case in range(cases):
    cname = ("p256", "p384", "k256")[case % 3]
    c = M.CURVES[cname]
    cv = ctx.curve(cname)
    nb = cv.nb
    pick = case % 7
    n = int({0: 1, 1: 2, 2: 63, 3: rng.integers(3, 5000), 4: lanes - 1, 5: lanes + 1 + int(rng.integers(0, 4000)), 6: rng.integers(5000, 300000)}[pick])
    u8 = dict(dtype=torch.uint8, device="cuda")
    d_s = torch.empty((n, nb), **u8); d_p = torch.empty((n, 2 * nb), **u8)
    first = int(rng.integers(0, 1 << 40))
    cv.synth_scalars_device(d_s, n, synth.SEED, first); cv.synth_points_device(d_p, n, synth.SEED, first)
    ctx.synchronize()
    edges = [0, 1, 2, c.n - 1, c.n - 2, c.n - 6, (c.n - 1) // 2, (c.n + 1) // 2, c.n, c.n + 7, (1 << (8 * nb)) - 1, (1 << 128) - 1, (1 << 128) + 1, 15, 16, 17]
    for v in edges:
        i = int(rng.integers(0, n))
        d_s[i] = torch.from_numpy(np.frombuffer(int(v).to_bytes(nb, "big"), dtype=np.uint8).copy()).cuda()
    for _ in range(3):
        d_p[int(rng.integers(0, n))] = 0                              # identity inputs
    if n > 4:
        d_p[n // 2] = d_p[n // 2 - 1]                                  # repeated point
    torch.cuda.synchronize()
    outs = {}
    for name, fl in (("ct", ecgpu.SECRET_SCALARS), ("ref", ecgpu.EXACT_REFERENCE), ("fast", 0)):
        o = torch.empty((n, 2 * nb), **u8); f = torch.empty((n,), **u8)
        cv.mul_device(d_s, d_p, o, n, d_out_inf=f, flags=fl)
        outs[name] = (o, f)
    # projective input (x z : y z : z) through the secret-scalar kernel
    xyz = torch.empty((n, 3 * nb), **u8)
    cv.mul_device(torch.from_numpy(np.tile(np.frombuffer((1).to_bytes(nb, "big"), dtype=np.uint8), (n, 1)).copy()).cuda(), d_p, xyz, n,
                  out_format=ecgpu.PROJECTIVE, flags=ecgpu.EXACT_REFERENCE)           # 1 * P in the reference's projective form
    o = torch.empty((n, 2 * nb), **u8); f = torch.empty((n,), **u8)
    cv.mul_device(d_s, xyz, o, n, point_format=ecgpu.PROJECTIVE, d_out_inf=f, flags=ecgpu.SECRET_SCALARS)
    outs["ct_proj"] = (o, f)
    sh = torch.empty((n, nb), **u8); ok = torch.empty((n,), **u8)
    cv.ecdh_device(d_s, d_p, sh, ok, n)
    ctx.synchronize()
    good = all(torch.equal(outs[k][0], outs["ref"][0]) and torch.equal(outs[k][1], outs["ref"][1]) for k in ("ct", "fast", "ct_proj"))
    valid = ok.bool()
    good = good and torch.equal(sh[valid], outs["ref"][0][valid][:, :nb]) and not bool(sh[~valid].any())
    # ok must be 0 exactly where the scalar is 0 or >= n or the point is the identity
    s_np = d_s.cpu().numpy(); p_np = d_p.cpu().numpy()
    ks = [int.from_bytes(bytes(r), "big") for r in s_np[: min(n, 3000)]]
    exp_ok = np.array([1 if (0 < k < c.n and p_np[i].any()) else 0 for i, k in enumerate(ks)], dtype=np.uint8)
    good = good and bool((ok.cpu().numpy()[: len(ks)] == exp_ok).all())
    bad += 0 if good else 1
    print("case %3d %s n=%7d %s" % (case, cname, n, "ok" if good else "MISMATCH"), flush=True)
    del d_s, d_p, outs, xyz, o, f, sh, ok
    torch.cuda.empty_cache()
ctx.close()
print("mismatches: %d" % bad)
sys.exit(1 if bad else 0)
