"""Why do two contexts on one card finish 2 x 2^24 k256 multiplications faster than one context finishes them in sequence?
Probe: (a) one context, 2^24; (b) one context, 2^25 in one call; (c) two contexts x 2^24 launched together (device group [0, 0]);
(d) two contexts, 2^23 each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import torch, ecgpu
from oracle import synth

def inputs(cv, n, first):
    s = torch.empty((n, 32), dtype=torch.uint8, device="cuda"); p = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(s, n, synth.SEED, first); cv.synth_points_device(p, n, synth.SEED, first)
    return s, p, torch.empty((n, 64), dtype=torch.uint8, device="cuda"), torch.empty((n,), dtype=torch.uint8, device="cuda")

ctx = ecgpu.Context(0); cv = ctx.curve("k256")
for lg in (24, 25):
    n = 1 << lg
    s, p, o, f = inputs(cv, n, 0); ctx.synchronize()
    for rep in range(3):
        ctx.timer_start(); cv.mul_device(s, p, o, n, d_out_inf=f); ms = ctx.timer_stop()
        print(f"one context, 2^{lg}: {ms:8.2f} ms = {ms / (n >> 24):8.2f} ms per 2^24", flush=True)
    del s, p, o, f
ctx.close()
for lg in (24, 23):
    n = 1 << lg
    g = ecgpu.Group([0, 0])
    bufs = [inputs(g.context(i).curve("k256"), n, i * n) for i in range(2)]
    g.synchronize()
    for rep in range(4):
        t0 = time.perf_counter()
        g.lincomb_sharded("k256", [b[0] for b in bufs], [b[1] for b in bufs], [b[2] for b in bufs], [n, n], d_out_inf=[b[3] for b in bufs])
        g.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        print(f"two contexts x 2^{lg}: {dt:8.2f} ms = {dt / (2 * n >> 24):8.2f} ms per 2^24", flush=True)
    g.close(); del bufs
