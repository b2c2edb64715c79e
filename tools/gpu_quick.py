"""Quick on-box timing of the k256 kernels (device-resident inputs). Usage: python tools/gpu_quick.py [log2n]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import torch, ecgpu
from oracle import synth
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
only = sys.argv[2] if len(sys.argv) > 2 else ""
n = 1 << lg
ctx = ecgpu.Context(0); cv = ctx.curve("k256")
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
d_s = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
d_p = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
d_o = torch.empty((n, 96), dtype=torch.uint8, device="cuda")
d_i = torch.empty((n,), dtype=torch.uint8, device="cuda")
cv.synth_scalars_device(d_s, n, synth.SEED); cv.synth_points_device(d_p, n, synth.SEED); ctx.synchronize()
for name, flags, fmt in (("mul exact-ref affine-out", ecgpu.EXACT_REFERENCE, ecgpu.AFFINE), ("mul fast affine-out", 0, ecgpu.AFFINE)):
    if only and only not in name:
        continue
    for rep in range(3):
        ctx.timer_start()
        cv.mul_device(d_s, d_p, d_o, n, out_format=fmt, d_out_inf=d_i, flags=flags)
        ms = ctx.timer_stop()
        print(f"{name}: n=2^{lg} {ms:.2f} ms  {n/ms*1e3/1e6:.2f} M scalar-mul/s", flush=True)
