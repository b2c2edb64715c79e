"""Quick on-box timing through the C ABI (device-resident inputs).
Usage: python tools/gpu_quick.py <curve> <log2n> [fixed|var] [ref]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import torch, ecgpu
from oracle import synth
cn = sys.argv[1] if len(sys.argv) > 1 else "k256"
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = sys.argv[3] if len(sys.argv) > 3 else "var"
flags = ecgpu.EXACT_REFERENCE if (len(sys.argv) > 4 and sys.argv[4] == "ref") else 0
n = 1 << lg
ctx = ecgpu.Context(0); cv = ctx.curve(cn); nb = cv.nb
from _opts import apply_env_options
apply_env_options(ctx)        # ECGPU_MSM_CBITS, ECGPU_FB_WINDOW ... -> ecgpu_set_option
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
d_s = torch.empty((n, nb), dtype=torch.uint8, device="cuda")
d_p = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
d_o = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda")
d_i = torch.empty((n,), dtype=torch.uint8, device="cuda")
cv.synth_scalars_device(d_s, n, synth.SEED)
if mode == "var":
    cv.synth_points_device(d_p, n, synth.SEED)
ctx.synchronize()
if mode == "lincomb2":
    d_s2 = torch.empty((2 * n, nb), dtype=torch.uint8, device="cuda"); d_p2 = torch.empty((2 * n, 2 * nb), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_s2, 2 * n, synth.SEED); cv.synth_points_device(d_p2, 2 * n, synth.SEED); ctx.synchronize()
    import ctypes
    for rep in range(3):
        ctx.timer_start()
        ctx.check(ctx.lib.ecgpu_lincomb_batch(ctx.handle, cv.id, ctypes.c_void_p(d_s2.data_ptr()), ctypes.c_void_p(d_p2.data_ptr()), 0, 2,
                                              ctypes.c_void_p(d_o.data_ptr()), 0, ctypes.c_void_p(d_i.data_ptr()), n, 1, flags))
        ms = ctx.timer_stop()
        print(f"{cn} lincomb2 {'ref' if flags else 'default'}: n=2^{lg} {ms:.2f} ms  {n/ms*1e3/1e6:.3f} M lincombs/s", flush=True)
    sys.exit(0)
if mode == "host":
    # the same call with HOST buffers: staging copies (pageable memory) + kernel + copy back, end to end
    import time, numpy as np
    hs, hp = d_s.cpu().numpy(), None
    cv.synth_points_device(d_p, n, synth.SEED); ctx.synchronize()
    hp = d_p.cpu().numpy()
    for rep in range(3):
        t0 = time.perf_counter()
        out, inf = cv.mul(hs, hp)
        dt = time.perf_counter() - t0
        print(f"{cn} var HOST buffers: n=2^{lg} {dt*1e3:.1f} ms  {n/dt/1e6:.2f} M scalar-mul/s  ({n*(3*nb+2*nb+1)/dt/1e9:.2f} GB/s over PCIe)", flush=True)
    ps, pp = ctx.pinned_array((n, nb)), ctx.pinned_array((n, 2 * nb))
    po, pi = ctx.pinned_array((n, 2 * nb)), ctx.pinned_array((n,))
    ps[:] = hs; pp[:] = hp
    for rep in range(3):
        t0 = time.perf_counter()
        cv.mul(ps, pp, out=po, out_inf=pi)
        dt = time.perf_counter() - t0
        print(f"{cn} var PINNED host buffers: n=2^{lg} {dt*1e3:.1f} ms  {n/dt/1e6:.2f} M scalar-mul/s  ({n*(3*nb+2*nb+1)/dt/1e9:.2f} GB/s over PCIe)", flush=True)
    assert bytes(po) == bytes(out)
    sys.exit(0)
if mode == "ecdsa":
    # sign n prehashes on the device, then time verification of the valid batch
    d_k = torch.empty((n, nb), dtype=torch.uint8, device="cuda"); d_z = torch.empty((n, nb), dtype=torch.uint8, device="cuda")
    cv.synth_scalars_device(d_k, n, synth.SEED, n); cv.synth_scalars_device(d_z, n, synth.SEED, 2 * n)
    d_sig = torch.empty((n, 2 * nb), dtype=torch.uint8, device="cuda"); d_rec = torch.empty((n,), dtype=torch.uint8, device="cuda")
    cv.mul_device(d_s, None, d_p, n)
    for rep in range(3):
        ctx.timer_start()
        cv.ecdsa_sign_device(d_s, d_k, d_z, d_sig, d_rec, d_i, n)
        ms = ctx.timer_stop()
        print(f"{cn} ecdsa sign: n=2^{lg} {ms:.2f} ms  {n/ms*1e3/1e6:.3f} M signatures/s", flush=True)
    for rep in range(3):
        ctx.timer_start()
        cv.ecdsa_verify_device(d_z, d_sig, d_p, d_i, n)
        ms = ctx.timer_stop()
        print(f"{cn} ecdsa verify: n=2^{lg} {ms:.2f} ms  {n/ms*1e3/1e6:.3f} M verifications/s  all_ok={bool(d_i.all())}", flush=True)
    sys.exit(0)
if mode == "msm":
    cv.synth_points_device(d_p, n, synth.SEED); ctx.synchronize()
    d_r = torch.empty((2 * nb,), dtype=torch.uint8, device="cuda")
    for rep in range(4):
        ctx.timer_start()
        cv.msm_device(d_s, d_p, n, d_r)
        ms = ctx.timer_stop()
        print(f"{cn} msm: n=2^{lg} {ms:.2f} ms  {n/ms*1e3/1e6:.2f} M points/s", flush=True)
    sys.exit(0)
for rep in range(3):
    ctx.timer_start()
    cv.mul_device(d_s, d_p if mode == "var" else None, d_o, n, d_out_inf=d_i, flags=flags)
    ms = ctx.timer_stop()
    print(f"{cn} {mode} {'ref' if flags else 'default'}: n=2^{lg} {ms:.2f} ms  {n/ms*1e3/1e6:.3f} M scalar-mul/s", flush=True)
