"""DIAGNOSTIC (libecgpu_bt<k>.so: -DK256_BLOCK_TIMES): when did every workgroup of the headline kernel start and end, and on which XCD / SE / CU?
Usage: ECGPU_LIB=.../lib_exp/libecgpu_bt4.so python tools/block_times_probe.py <workgroups per CU>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rustcrypto-elliptic-curves_amd"))
import numpy as np, torch, ecgpu
from oracle import synth
per_cu = int(sys.argv[1])
n = 1 << 24
ctx = ecgpu.Context(0); cv = ctx.curve("k256")
s = torch.empty((n, 32), dtype=torch.uint8, device="cuda"); p = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
o = torch.empty((n, 64), dtype=torch.uint8, device="cuda"); f = torch.empty((n,), dtype=torch.uint8, device="cuda")
cv.synth_scalars_device(s, n, synth.SEED, 0); cv.synth_points_device(p, n, synth.SEED, 0); ctx.synchronize()
for rep in range(2):
    ctx.timer_start(); cv.mul_device(s, p, o, n, d_out_inf=f); ms = ctx.timer_stop()
grid = 256 * per_cu
ws = ctx.debug_workspace(0)
tail = np.frombuffer(ws[grid * 256 * 2048: grid * 256 * 2048 + grid * 32], dtype=np.uint64).reshape(grid, 4)
t0 = tail[:, 0].min()
st = (tail[:, 0] - t0) / 1e5            # ms (100 MHz clock)
en = (tail[:, 1] - t0) / 1e5
hw = tail[:, 2].astype(np.int64); xcc = (tail[:, 3].astype(np.int64)) & 15
cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print(f"kernel {ms:.2f} ms, {grid} workgroups; start: min {st.min():.2f} median {np.median(st):.2f} max {st.max():.2f} ms; end: min {en.min():.2f} median {np.median(en):.2f} max {en.max():.2f} ms; duration: min {(en-st).min():.2f} median {np.median(en-st):.2f} max {(en-st).max():.2f}")
print("workgroups started in the first millisecond:", int((st < 1.0).sum()))
for x in range(8):
    m = xcc == x
    print(f"  XCC {x}: {int(m.sum()):5d} workgroups, started at once {int((m & (st < 1.0)).sum()):5d}, last end {en[m].max():8.2f} ms, mean duration {np.mean((en - st)[m]):8.2f} ms, distinct (se, sh, cu) {len(set(zip(se[m], sh[m], cu[m])))}")
# per CU: how many workgroups started at once
key = xcc * 4096 + se * 256 + sh * 16 + cu
first = st < 1.0
vals, counts = np.unique(key[first], return_counts=True)
print("CUs that hold workgroups from the start:", len(vals), "; workgroups per such CU: min", counts.min(), "max", counts.max(), "histogram", np.bincount(counts).tolist())
