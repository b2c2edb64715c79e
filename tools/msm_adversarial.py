"""Timing and correctness of the MSM on adversarial inputs (equal scalars: whole windows land in one bucket / coarse bin).
    python tools/msm_adversarial.py [log2n]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import numpy as np
import torch
import ecgpu
from oracle import ecmodel as M, synth

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
ctx = ecgpu.Context(0)
ctx.set_option(ecgpu.OPT_MSM_SMALL_PATH, 0)
cv = ctx.curve("k256")
c = M.K256
d_p = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
d_o = torch.empty((64,), dtype=torch.uint8, device="cuda")
d_o2 = torch.empty((64,), dtype=torch.uint8, device="cuda")
cv.synth_points_device(d_p, n, synth.SEED, 0)
ctx.synchronize()
for name, k in (("all ones (a plain sum of points)", 1), ("all n - 1", c.n - 1), ("one repeated 256-bit scalar", synth.scalar(c, 5))):
    s = np.tile(np.frombuffer(k.to_bytes(32, "big"), dtype=np.uint8), (n, 1))
    d_s = torch.from_numpy(s.copy()).cuda()
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        cv.msm_device(d_s, d_p, n, d_o)
        ctx.synchronize()
        dt = time.perf_counter() - t0
    # check: k * (sum of points) computed as the sum with unit scalars (the term-by-term path), then one multiplication
    ctx.set_option(ecgpu.OPT_MSM_SMALL_PATH, 1)
    ones = torch.zeros((n, 32), dtype=torch.uint8, device="cuda"); ones[:, 31] = 1
    parts = []
    for lo in range(0, n, 1 << 16):                      # chunks below the small-path threshold
        hi = min(n, lo + (1 << 16))
        t = torch.empty((64,), dtype=torch.uint8, device="cuda")
        cv.msm_device(ones[lo:hi], d_p[lo:hi], hi - lo, t)
        parts.append(t)
    ctx.synchronize()
    allp = torch.stack(parts)
    tot = torch.empty((64,), dtype=torch.uint8, device="cuda")
    cv.msm_device(ones[:len(parts)], allp, len(parts), tot)
    ctx.synchronize()
    ctx.set_option(ecgpu.OPT_MSM_SMALL_PATH, 0)
    want, _ = cv.mul(np.frombuffer(k.to_bytes(32, "big"), dtype=np.uint8).reshape(1, 32), tot.cpu().numpy().reshape(1, 64))
    ok = bytes(want[0]) == bytes(d_o.cpu().numpy())
    print(f"{name}: n=2^{lg} {dt*1e3:.1f} ms  correct={ok}", flush=True)
ctx.close()
