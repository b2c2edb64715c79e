#!/usr/bin/env python3
"""Counter evidence that the reference (constant-time) schedules execute the same instructions whatever the scalar is.

Run under rocprofv3 (counters in their own pass, no traces):
    rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/ct -- python3 tools/ct_evidence.py
Each kernel below is launched three times on 2^16 scalars: all equal to 1 (every digit but one is zero), all equal to
n - 1 / 0x8888.. patterns (every digit non-zero), and random.  tools/ct_summarize.py then checks that the per-dispatch
counters of the three launches are identical for the reference schedule (and shows that the throughput schedule's differ).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import numpy as np
import torch
import ecgpu
from oracle import ecmodel as M

n = 1 << 16
ctx = ecgpu.Context(0)
for cname in ("k256", "p256", "p384"):
    c = M.CURVES[cname]
    cv = ctx.curve(cname)
    nb = cv.nb
    rng = np.random.default_rng(7)
    sets = []
    sets.append(np.tile(np.frombuffer((1).to_bytes(nb, "big"), dtype=np.uint8), (n, 1)))
    sets.append(np.tile(np.frombuffer((int("7" * (2 * nb), 16) % c.n).to_bytes(nb, "big"), dtype=np.uint8), (n, 1)))
    r = rng.integers(0, 256, size=(n, nb), dtype=np.uint8)
    r[:, 0] &= 0x7F
    sets.append(r)
    d_o = torch.empty((n, 3 * nb), dtype=torch.uint8, device="cuda")
    G = np.tile(np.frombuffer(M.i2b(c, c.gx) + M.i2b(c, c.gy), dtype=np.uint8), (n, 1))
    d_p = torch.from_numpy(G.copy()).cuda()
    for flags in (ecgpu.EXACT_REFERENCE, 0):
        for s in sets:
            d_s = torch.from_numpy(np.ascontiguousarray(s)).cuda()
            torch.cuda.synchronize()
            cv.mul_device(d_s, None, d_o, n, out_format=ecgpu.PROJECTIVE, flags=flags)        # mul_by_generator
            ctx.synchronize()
            cv.mul_device(d_s, d_p, d_o, n, out_format=ecgpu.PROJECTIVE, flags=flags)         # variable base
            ctx.synchronize()
    # ECDH: a secret scalar on a variable base (ECGPU_SECRET_SCALARS: vbct::mul_kernel on P-256 / P-384, the reference schedule
    # on secp256k1), random points, the three scalar sets plus one more: n - 2 / n - 6, where a schedule without the fold breaks
    d_q = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
    cv.synth_points_device(d_q, n, 0xEC5CA1A5, 0)
    ctx.synchronize()
    extra = np.tile(np.frombuffer((c.n - (2 if cname != "p384" else 6)).to_bytes(nb, "big"), dtype=np.uint8), (n, 1))
    for s in sets + [extra]:
        d_s = torch.from_numpy(np.ascontiguousarray(s)).cuda()
        torch.cuda.synchronize()
        cv.mul_device(d_s, d_q, d_o, n, flags=ecgpu.SECRET_SCALARS)
        ctx.synchronize()
    # signing: the nonce k takes the three sets (default flags: the constant-time fixed-base kernel fb::mul_ct_kernel)
    d_d = torch.from_numpy(np.ascontiguousarray(sets[2])).cuda()
    d_z = torch.from_numpy(np.ascontiguousarray(sets[2][::-1])).cuda()
    d_sig = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
    d_rec = torch.empty((n,), dtype=torch.uint8, device="cuda")
    d_ok = torch.empty((n,), dtype=torch.uint8, device="cuda")
    for s in sets:
        d_k = torch.from_numpy(np.ascontiguousarray(s)).cuda()
        torch.cuda.synchronize()
        cv.ecdsa_sign_device(d_d, d_k, d_z, d_sig, d_rec, d_ok, n)
        ctx.synchronize()
ctx.close()
print("done")
