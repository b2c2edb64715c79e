#!/bin/bash
# MSM experiments on the GPU box: the MSM tests, timing of the 2^23-term sum for several settings of the environment
# switches (window width ECGPU_MSM_CBITS, bucket-sum runs per lane ECGPU_MSM_ROUNDS), then a kernel trace of the default.
#   gpurun --timeout 900 -- 'bash tools/msm_sweep.sh'        (SWEEP="19:4 16:4" picks CBITS:ROUNDS pairs; CURVE, LG)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/msm_sweep
rm -rf "$OUT"; mkdir -p "$OUT"
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 800 python -m pytest tests/test_gpu_msm.py -x -q > "$OUT/tests.log" 2>&1 || { tail -30 "$OUT/tests.log"; exit 1; }
  tail -2 "$OUT/tests.log"
fi
for cfg in ${SWEEP:-19:4 19:1 19:2 19:8 16:4 16:1 16:8}; do
  echo "== ECGPU_MSM_CBITS:ECGPU_MSM_ROUNDS = $cfg"
  ECGPU_MSM_CBITS=${cfg%%:*} ECGPU_MSM_ROUNDS=${cfg##*:} timeout -k 10 120 python tools/gpu_quick.py ${CURVE:-k256} ${LG:-23} msm 2>&1 | grep msm | tail -2
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/gpu_quick.py ${CURVE:-k256} ${LG:-23} msm > "$OUT/trace.log" 2>&1
find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
python3 tools/msm_timeline.py "$OUT/trace" > "$OUT/timeline.txt" 2>&1
cat "$OUT/timeline.txt"
