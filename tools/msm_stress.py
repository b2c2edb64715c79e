#!/usr/bin/env python3
"""Randomised cross-check of the MSM paths on the GPU box: random sizes and input mixes (duplicate points, zero and
repeated scalars, identity points, clustered scalars, tiny slabs, odd bucket-sum run counts), every case computed by the
term-by-term path (chunks below its threshold, folded with complete additions) and by the bucket method with 16- and
19-bit windows; all three must give the same bytes.     python tools/msm_stress.py [cases] [seed]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import numpy as np
import torch
import ecgpu
from ecgpu import parallel
from oracle import synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = ecgpu.Context(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx.set_stream(st.cuda_stream)
bad = 0
for case in range(cases):
    cname = ["k256", "p256", "p384"][case % 3] if case % 4 else "k256"
    cv = ctx.curve(cname)
    nb = cv.nb
    n = int(rng.choice([rng.integers(1, 70), rng.integers(70, 5000), rng.integers(5000, 90000), rng.integers(90000, 400000)]))
    d_s = torch.empty((n, nb), dtype=torch.uint8, device="cuda")
    d_p = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_s, n, 1000 + case)
    cv.synth_points_device(d_p, n, 2000 + case)
    ctx.synchronize()
    mix = int(rng.integers(0, 6))
    if mix == 1 and n > 4:                       # clustered scalars: a few values only
        k = int(rng.integers(1, 5))
        d_s[:] = d_s[torch.from_numpy(rng.integers(0, k, size=n)).cuda()]
    elif mix == 2 and n > 4:                     # duplicates of points and of (scalar, point) pairs, zeros, identities
        idx = torch.from_numpy(rng.integers(0, max(1, n // 3), size=n)).cuda()
        d_p[:] = d_p[idx]
        d_s[::7] = 0
        d_p[3::11] = 0
    elif mix == 3:                               # small scalars (most windows empty)
        d_s[:, : nb - 3] = 0
    elif mix == 4 and n > 2:                     # P and -P with equal scalars, pairs cancel
        h = n // 2
        d_p[h:2 * h] = d_p[:h]
        d_s[h:2 * h] = d_s[:h]
    torch.cuda.synchronize()
    res = {}
    for path in ("term", "b16", "b19"):
        ctx.set_option(ecgpu.OPT_MSM_SMALL_PATH, 1); ctx.set_option(ecgpu.OPT_MSM_WINDOW_BITS, 0); ctx.set_option(ecgpu.OPT_MSM_SLAB_TERMS, 0)
        out = torch.empty((2 * nb,), dtype=torch.uint8, device="cuda")
        if path == "term":
            step = 60000
            parts = []
            for lo in range(0, n, step):
                hi = min(n, lo + step)
                t = torch.empty((3 * nb,), dtype=torch.uint8, device="cuda")
                cv.msm_device(d_s[lo:hi], d_p[lo:hi], hi - lo, t, out_format=ecgpu.PROJECTIVE)
                parts.append(t)
            allp = torch.stack(parts).contiguous()
            parallel.fold_points_device(cv, allp, len(parts), torch.empty_like(allp), out, torch.empty((1,), dtype=torch.uint8, device="cuda"))
        else:
            ctx.set_option(ecgpu.OPT_MSM_SMALL_PATH, 0)
            ctx.set_option(ecgpu.OPT_MSM_WINDOW_BITS, int(path[1:]))
            if case % 5 == 0 and n > 3000:
                ctx.set_option(ecgpu.OPT_MSM_SLAB_TERMS, int(rng.integers(1024, max(1025, n // 2))))
            cv.msm_device(d_s, d_p, n, out)
        ctx.synchronize()
        res[path] = bytes(out.cpu().numpy())
    ok = res["term"] == res["b16"] == res["b19"]
    bad += 0 if ok else 1
    print("case %2d %s n=%6d mix=%d %s" % (case, cname, n, mix, "ok" if ok else "MISMATCH"), flush=True)
ctx.close()
print("mismatches: %d" % bad)
sys.exit(1 if bad else 0)
