#!/bin/bash
# Copy what tools/round_evidence.sh A and B left under gpurun_out/ into profiles/ (tracked).   bash tools/collect_evidence.sh [r04]
cd "$(dirname "$0")/.."
R=${1:-r04}
cp gpurun_out/pmc_summary_${R}_*.json profiles/
cp gpurun_out/${R}_*_kernel_stats.csv profiles/
for f in ct_counters msm_sizes k256_msm_2p23_timeline secondary_entry_points ct_varbase_bench host_pipeline; do
  [ -f gpurun_out/$R/$f.txt ] && grep -v "amdgpu.ids" gpurun_out/$R/$f.txt > profiles/${R}_$f.txt
done
[ -f gpurun_out/$R/bench_k256_msm_forcedist.json ] && cp gpurun_out/$R/bench_k256_msm_forcedist.json profiles/${R}_bench_k256_msm_forcedist.json
python3 tools/kernel_resources.py > profiles/${R}_kernel_resources.txt
sed -i '1i # Secret-scalar variable-base multiplication (ECDH) at 2^22 device-resident units (tools/ct_varbase_bench.py 22): the constant-time kernel (ct), the reference schedule (ref) and the public-data throughput schedule (fast); best of 3 / 2 / 3' profiles/${R}_ct_varbase_bench.txt
