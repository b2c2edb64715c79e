"""On-box A/B of the k256 fast-kernel occupancy variants (interleaved rounds in one process per variant)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lg = sys.argv[1] if len(sys.argv) > 1 else "22"
for rnd in range(2):
    for w in ("2", "3", "4"):
        env = dict(os.environ, ECGPU_K256_FAST_WAVES=w)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_quick.py"), lg, "fast"], env=env, capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if "fast" in l]
        print("waves", w, "|", lines[-1] if lines else r.stderr[-300:], flush=True)
