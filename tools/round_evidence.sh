#!/bin/bash
# Everything profiles/ holds for a round, in two GPU calls (each well under the 20-minute limit):
#   gpurun --timeout 1200 -- 'GIT_COMMIT=<sha> ROUND=r04 bash tools/round_evidence.sh A'
#   gpurun --timeout 1200 -- 'GIT_COMMIT=<sha> ROUND=r04 bash tools/round_evidence.sh B'
# A: kernel trace + PMC passes of the headline and of config 5, the constant-time counter evidence.
# B: the same for configs 3 and 4, the MSM timeline and sizes, the secondary entry points, the forced-RCCL bench line.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
ROUND=${ROUND:-r04}
O=gpurun_out/$ROUND
mkdir -p "$O"
part=${1:-A}
prof() { WORKLOAD=$1 bash tools/profile_workload.sh > "$O/prof_$1.log" 2>&1; echo "profile $1 rc=$?"; rm -rf gpurun_out/prof_$1; }
if [ "$part" = A ]; then
  prof k256_varbase
  prof p384_varbase
  timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/ct -- python3 tools/ct_evidence.py > "$O/ct.log" 2>&1; echo "ct rc=$?"
  python3 tools/ct_summarize.py gpurun_out/ct > "$O/ct_counters.txt"; rm -rf gpurun_out/ct
  grep -c "counters identical: True" "$O/ct_counters.txt" || true
else
  prof p256_fixedbase
  prof k256_msm
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/msm_trace -- python3 bench.py --workload k256_msm --steps 3 --warmup 1 --no-cpu-baseline > "$O/msm_trace.log" 2>&1
  python3 tools/msm_timeline.py gpurun_out/msm_trace > "$O/k256_msm_2p23_timeline.txt" 2>&1; rm -rf gpurun_out/msm_trace
  tail -3 "$O/k256_msm_2p23_timeline.txt"
  { echo "# k256 MSM time against the number of terms (tools/gpu_quick.py <curve> <log2n> msm, library's own choice of window width; 2^24 and up: 16-bit windows in slabs of 2^24 terms)";
    for lg in 20 21 22 23 24 25 26; do timeout -k 10 200 python tools/gpu_quick.py k256 $lg msm 2>&1 | grep msm | tail -1; done;
    for lg in 22 23; do timeout -k 10 200 python tools/gpu_quick.py p256 $lg msm 2>&1 | grep msm | tail -1; done;
    timeout -k 10 200 python tools/gpu_quick.py p384 22 msm 2>&1 | grep msm | tail -1; } > "$O/msm_sizes.txt" 2>&1
  cat "$O/msm_sizes.txt"
  timeout -k 10 600 python tools/util_bench.py 20 > "$O/secondary_entry_points.txt" 2>&1; echo "util rc=$?"
  { echo "# linear combinations of 3 .. 1024 terms (tools/lincomb_bench.py 22): shared doublings (csrc/straus.hpp) against the term-by-term form of rounds 1-3"; timeout -k 10 400 python tools/lincomb_bench.py 22 2>&1 | grep lincomb; } >> "$O/secondary_entry_points.txt"
  timeout -k 10 300 python tools/host_pipeline_bench.py k256 24 4 msm > "$O/host_pipeline.txt" 2>&1; echo "hostpipe rc=$?"
  timeout -k 10 300 python bench.py --workload k256_msm --force-dist --steps 5 --no-cpu-baseline > "$O/bench_k256_msm_forcedist.json" 2> "$O/bench_k256_msm_forcedist.err"; echo "forcedist rc=$?"
  timeout -k 10 300 python tools/ct_varbase_bench.py 22 > "$O/ct_varbase_bench.txt" 2>&1; echo "ctbench rc=$?"
fi
