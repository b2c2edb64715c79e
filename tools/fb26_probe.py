import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import torch, ecgpu
from oracle import synth
ctx = ecgpu.Context(0); cv = ctx.curve("p256")
n = 1 << 24
d_s = torch.empty((n, 32), dtype=torch.uint8, device="cuda"); d_o = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
cv.synth_scalars_device(d_s, n, synth.SEED); ctx.synchronize()
t0 = time.perf_counter(); x = torch.empty((54 << 30,), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1 = time.perf_counter()
del x; torch.cuda.empty_cache(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("torch alloc 54 GB %.3f s, free %.3f s" % (t1 - t0, t2 - t1))
for rep in range(3):
    t0 = time.perf_counter(); cv.mul_device(d_s, None, d_o, n); ctx.synchronize(); print("call %d: %.1f ms" % (rep, (time.perf_counter() - t0) * 1e3), flush=True)
