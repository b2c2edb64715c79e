#!/bin/bash
# MSM experiments on the GPU box: the MSM tests, timing of the 2^23-term sum for several settings of the environment
# switches, then a kernel trace of the default.
#   gpurun --timeout 900 -- 'bash tools/msm_sweep.sh'        (SWEEP="8:64 4:64" picks ECGPU_MSM_SPLIT:ECGPU_MSM_WGS pairs)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/msm_sweep
rm -rf "$OUT"; mkdir -p "$OUT"
python -m pytest tests/test_gpu_msm.py -x -q > "$OUT/tests.log" 2>&1 || { tail -20 "$OUT/tests.log"; exit 1; }
tail -2 "$OUT/tests.log"
for cfg in ${SWEEP:-8:64 8:4 4:64 16:64 2:64 1:64}; do
  echo "== ECGPU_MSM_SPLIT:ECGPU_MSM_WGS = $cfg"
  ECGPU_MSM_SPLIT=${cfg%%:*} ECGPU_MSM_WGS=${cfg##*:} python tools/gpu_quick.py k256 23 msm 2>&1 | grep msm | tail -2
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/gpu_quick.py k256 23 msm > "$OUT/trace.log" 2>&1
find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
python3 - <<'EOF'
import csv
rows = list(csv.DictReader(open("gpurun_out/msm_sweep/kernel_stats.csv")))
for r in rows[:24]:
    print("%-60s calls %3s  avg %10.1f us  %5s %%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
EOF
