"""Secret-scalar variable-base multiplication (ECDH) on device-resident batches: the constant-time kernel (ECGPU_SECRET_SCALARS),
the reference schedule (ECGPU_EXACT_REFERENCE) and the public-data throughput schedule side by side.
python tools/ct_varbase_bench.py [log2n] [curves]  -> one line per (curve, schedule)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import torch
import ecgpu
from oracle import synth

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 21
curves = sys.argv[2].split(",") if len(sys.argv) > 2 else ["p256", "p384", "k256"]
n = 1 << lg
ctx = ecgpu.Context(0)
for cn in curves:
    cv = ctx.curve(cn)
    nb = cv.nb
    u8 = dict(dtype=torch.uint8, device="cuda")
    s = torch.empty((n, nb), **u8); p = torch.empty((n, 2 * nb), **u8)
    o = {k: torch.empty((n, 2 * nb), **u8) for k in ("ct", "ref", "fast")}
    f = torch.empty((n,), **u8)
    cv.synth_scalars_device(s, n, synth.SEED, 0); cv.synth_points_device(p, n, synth.SEED, 0)
    ctx.synchronize()
    res = {}
    for name, flags, reps in (("ct", ecgpu.SECRET_SCALARS, 3), ("ref", ecgpu.EXACT_REFERENCE, 2), ("fast", 0, 3)):
        best = 1e9
        for _ in range(reps):
            ctx.timer_start()
            cv.mul_device(s, p, o[name], n, d_out_inf=f, flags=flags)
            best = min(best, ctx.timer_stop())
        res[name] = best
        print(f"{cn:5s} {name:5s} n=2^{lg} {best:9.3f} ms {n / best / 1e3:9.2f} M/s", flush=True)
    assert torch.equal(o["ct"], o["ref"]) and torch.equal(o["ct"], o["fast"]), cn
    print(f"{cn:5s} secret-scalar kernel / reference schedule = {res['ref'] / res['ct']:.2f}x; / throughput schedule = {res['fast'] / res['ct']:.2f}x", flush=True)
    del s, p, o
    torch.cuda.empty_cache()
ctx.close()
