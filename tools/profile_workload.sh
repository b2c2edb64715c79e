#!/bin/bash
# Collect the evidence behind bench.py's roofline object for ONE workload of bench.py, on the GPU box:
#   gpurun --timeout 1100 -- 'GIT_COMMIT=<sha> ROUND=r03 WORKLOAD=k256_varbase bash tools/profile_workload.sh'
#   WORKLOAD = k256_varbase | p256_fixedbase | k256_msm | p384_varbase  (KERNEL_MATCH defaults to bench.py's pmc_match)
# 1. kernel trace + stats of the bench command, 2. PMC counters, one group per pass (never combined with traces),
# 3. tools/pmc_summarize.py folds them into gpurun_out/pmc_summary_${ROUND}_${WORKLOAD}.json (with the commit passed in and the
#    hash of the kernel sources), 4. the kernel-stats CSV next to it.  Copy both into profiles/ afterwards.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
WORKLOAD=${WORKLOAD:-k256_varbase}
ROUND=${ROUND:-r04}
OUT=gpurun_out/prof_$WORKLOAD
rm -rf "$OUT"; mkdir -p "$OUT"
MATCH=${KERNEL_MATCH:-$(python3 -c "import bench; print(bench.WORKLOADS['$WORKLOAD']['pmc_match'])")}
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-io --workload $WORKLOAD"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/trace.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY"; do
  tag=$(echo "$grp" | tr ' ' '+')
  echo "[pmc] $grp"
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$tag" -- python3 $ARGS > "$OUT/pmc_$tag.log" 2>&1
done
# known read volume of the headline kernel: 2^24 units x (50.4 table blocks of 64 B - 52 digits x 31/32, 5-bit windows since round 4 - + 96 B of input)
# (rounds 1-3, 4-bit windows: 61.9 blocks, 68073553920 bytes)
KNOWN=${KNOWN_READ_BYTES:--}
[ "$WORKLOAD" = k256_varbase ] && [ "$KNOWN" = "-" ] && KNOWN=55700357120
SUM=gpurun_out/pmc_summary_${ROUND}_${WORKLOAD}.json
python3 tools/pmc_summarize.py "$OUT" "$MATCH" "$KNOWN" "$WORKLOAD" "${GIT_COMMIT:-unknown}" > "$SUM"
cat "$SUM"
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" gpurun_out/${ROUND}_${WORKLOAD}_kernel_stats.csv
