#!/bin/bash
# Round-4 A/B measurements, set three: 5-bit windows in the public-data variable-base kernel of P-384 / P-256
# (csrc/varbase_lane.hpp WB = 5: 16 table entries, 77 / 52 windows instead of 97 / 65; make variant NAME=p384w5 TU=ops_p384 DEFS="-D'VB_WINDOW(C)=5'")
#   gpurun --timeout 900 -- 'mkdir -p gpurun_out/r4 && bash tools/ab_round4b.sh > gpurun_out/r4/ab_3.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
B="--no-cpu-baseline --no-other-configs --no-host-io --steps 4 --warmup 1"
line() { python -c 'import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("   ", d["config"]["workload"][:40], "ms_per_step", round(d["ms_per_step"], 3), "kernel_ms", round(d["roofline"]["kernel_ms"], 3), "value", "%.4g" % d["value"], "parity", d["parity_ok"])'; }
for rep in 1 2 3; do
  for v in default p384w5; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    echo "#### p384 variable base 2^22 (config 5), library: $v (pass $rep)"
    timeout -k 10 200 python bench.py $B --workload p384_varbase 2>/dev/null | line
  done
  for v in default p256w5; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    echo "#### p256 variable base 2^22, library: $v (pass $rep)"
    timeout -k 10 120 python tools/gpu_quick.py p256 22 var 2>&1 | grep "default:" | tail -2
  done
done
