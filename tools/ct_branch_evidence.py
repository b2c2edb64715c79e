#!/usr/bin/env python3
"""Counter evidence for the branch-free secp256k1 field operations of the constant-time kernels (csrc/ops_k256_ct.hip).

Run under rocprofv3 (counters in their own pass, no traces), once per build of the library:
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/ctb -- python3 tools/ct_branch_evidence.py
Every kernel is launched on 2^21 units with FOUR DIFFERENT RANDOM scalar sets (and the same points).  The carry paths that round 3
left as real branches fire with probability 2^-26 .. 2^-32 per field operation: ~1 700 multiplications x 2^21 units is ~50 firings per
launch, each of which makes its wave execute a few instructions more - invisible at 2^16 units (profiles/r03_ct_counters.txt), visible
here as SQ_INSTS_VALU / SALU counts that differ between the random sets.  In the branch-free build (ECGPU_K256_BRANCHFREE, the one
that ships) the counts must be IDENTICAL for all sets.  tools/ct_summarize.py folds the CSV."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import torch
import ecgpu

n = 1 << 21
ctx = ecgpu.Context(0)
cv = ctx.curve("k256")
u8 = dict(dtype=torch.uint8, device="cuda")
d_q = torch.empty((n, 64), **u8)
cv.synth_points_device(d_q, n, 0xEC5CA1A5, 0)
d_o = torch.empty((n, 64), **u8)
d_sig = torch.empty((n, 64), **u8); d_rec = torch.empty((n,), **u8); d_ok = torch.empty((n,), **u8)
sets = []
for seed in (101, 202, 303, 404):
    s = torch.empty((n, 32), **u8)
    cv.synth_scalars_device(s, n, seed, 0)
    sets.append(s)
ctx.synchronize()
for s in sets:                                   # ECDH kernel: k256_mul_ct_kernel
    cv.mul_device(s, d_q, d_o, n, flags=ecgpu.SECRET_SCALARS)
    ctx.synchronize()
for s in sets:                                   # signing: fb::mul_ct_kernel (the nonce takes the four sets)
    cv.ecdsa_sign_device(sets[0], s, sets[1], d_sig, d_rec, d_ok, n)
    ctx.synchronize()
m = 1 << 19
for s in sets:                                   # the reference schedule (exact X, Y, Z): lincomb_ref_kernel<CurveK256, 1>
    cv.mul_device(s[:m], d_q[:m], d_o[:m], m, flags=ecgpu.EXACT_REFERENCE)
    ctx.synchronize()
ctx.close()
print("done")
