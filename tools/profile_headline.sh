#!/bin/bash
# Collect the evidence behind bench.py's roofline object for the headline workload, on the GPU box:
#   gpurun --timeout 1100 -- 'bash tools/profile_headline.sh'
#   other workloads: TAG=p384 BENCH_ARGS="--workload p384_varbase" KERNEL_MATCH=vb::mul_kernel bash tools/profile_headline.sh
# 1. kernel trace + stats of the bench command, 2. PMC counters, one group per pass (never combined with
# traces), 3. tools/pmc_summarize.py folds them into gpurun_out/pmc_summary.json.
# Copy the summaries you want judged into profiles/ afterwards.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${TAG:-headline}
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs ${BENCH_ARGS:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/trace.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY"; do
  tag=$(echo "$grp" | tr ' ' '+')
  echo "[pmc] $grp"
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$tag" -- python3 $ARGS > "$OUT/pmc_$tag.log" 2>&1
done
# known read volume of the headline kernel: 2^24 units x (61.9 table blocks of 64 B + 96 B of input)
if [ "$TAG" = headline ]; then KNOWN=${KNOWN_READ_BYTES:-68073553920}; SUM=gpurun_out/pmc_summary.json; else KNOWN=${KNOWN_READ_BYTES:-}; SUM=gpurun_out/pmc_summary_$TAG.json; fi
python3 tools/pmc_summarize.py "$OUT" "${KERNEL_MATCH:-k256_mul_fast_kernel}" $KNOWN > "$SUM"
cat "$SUM"
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_kernel_stats.csv
[ "$TAG" = headline ] || exit 0
# constant-time evidence: per-dispatch instruction counters of the reference schedules on three very different scalar sets
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/ct -- python3 tools/ct_evidence.py > gpurun_out/ct.log 2>&1
python3 tools/ct_summarize.py gpurun_out/ct > gpurun_out/ct_counters.txt
cat gpurun_out/ct_counters.txt
