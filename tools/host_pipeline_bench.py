"""PCIe-inclusive rates of the host-buffer path (ECGPU_MEM_HOST, csrc/host_pipe.hpp) beside the device-resident rate.
Usage: python tools/host_pipeline_bench.py [curve] [log2n] [reps]   ->  text for profiles/r04_host_pipeline.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import numpy as np
import torch
import ecgpu
from oracle import synth

cn = sys.argv[1] if len(sys.argv) > 1 else "k256"
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 24
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
n = 1 << lg
ctx = ecgpu.Context(0)
cv = ctx.curve(cn)
nb = cv.nb
d_s = torch.empty((n, nb), dtype=torch.uint8, device="cuda")
d_p = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
d_o = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
d_i = torch.empty((n,), dtype=torch.uint8, device="cuda")
cv.synth_scalars_device(d_s, n, synth.SEED)
cv.synth_points_device(d_p, n, synth.SEED)
ctx.synchronize()
bytes_per_unit = 3 * nb + 2 * nb + 1
print(f"# {cn} variable base, n = 2^{lg}, {bytes_per_unit} B per unit over PCIe; host: {os.cpu_count()} cpus; chunks {ecgpu.host_chunk_schedule(n, 1 << 23 if cn == 'k256' else 1 << 22)}")
best = {}
for rep in range(reps):
    ctx.timer_start()
    cv.mul_device(d_s, d_p, d_o, n, d_out_inf=d_i)
    ms = ctx.timer_stop()
    best["device"] = min(best.get("device", 1e9), ms)
    print(f"device-resident: {ms:8.2f} ms  {n / ms * 1e3 / 1e6:7.2f} M/s", flush=True)
want = d_o.cpu().numpy()
hs, hp = d_s.cpu().numpy(), d_p.cpu().numpy()
out, inf = np.zeros((n, 2 * nb), dtype=np.uint8), np.zeros(n, dtype=np.uint8)
out[:] = 1; inf[:] = 1                         # touch the pages: a caller's buffers are not fresh from calloc
for rep in range(reps):
    t0 = time.perf_counter()
    cv.mul(hs, hp, out=out, out_inf=inf)
    dt = time.perf_counter() - t0
    best["pageable"] = min(best.get("pageable", 1e9), dt * 1e3)
    print(f"pageable host:   {dt * 1e3:8.2f} ms  {n / dt / 1e6:7.2f} M/s  {n * bytes_per_unit / dt / 1e9:6.2f} GB/s", flush=True)
assert bytes(out) == bytes(want)
ps, pp = ctx.pinned_array((n, nb)), ctx.pinned_array((n, 2 * nb))
po, pi = ctx.pinned_array((n, 2 * nb)), ctx.pinned_array((n,))
ps[:] = hs
pp[:] = hp
for rep in range(reps):
    t0 = time.perf_counter()
    cv.mul(ps, pp, out=po, out_inf=pi)
    dt = time.perf_counter() - t0
    best["pinned"] = min(best.get("pinned", 1e9), dt * 1e3)
    print(f"pinned host:     {dt * 1e3:8.2f} ms  {n / dt / 1e6:7.2f} M/s  {n * bytes_per_unit / dt / 1e9:6.2f} GB/s", flush=True)
assert bytes(po) == bytes(want)
print("best: " + "  ".join(f"{k} {n / v * 1e3 / 1e6:.2f} M/s ({v:.1f} ms)" for k, v in best.items()) + "  parity ok", flush=True)
if len(sys.argv) > 4 and sys.argv[4] == "msm":
    d_r = torch.empty((2 * nb,), dtype=torch.uint8, device="cuda")
    for rep in range(3):
        ctx.timer_start(); cv.msm_device(d_s, d_p, n, d_r); ms = ctx.timer_stop()
        print(f"msm device-resident: {ms:8.2f} ms  {n / ms * 1e3 / 1e6:7.1f} M points/s", flush=True)
    w = bytes(d_r.cpu().numpy())
    for name, a, b in (("pageable", hs, hp), ("pinned", ps, pp)):
        for rep in range(3):
            t0 = time.perf_counter(); r = cv.msm(a, b); dt = time.perf_counter() - t0
            print(f"msm {name:8s} host: {dt * 1e3:8.2f} ms  {n / dt / 1e6:7.1f} M points/s  {n * 3 * nb / dt / 1e9:6.2f} GB/s", flush=True)
        assert bytes(r) == w
