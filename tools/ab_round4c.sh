#!/bin/bash
# Round-4 A/B measurements, set four: signed 5-bit windows in the secp256k1 throughput schedule (csrc/mulfast_k256.hpp K256_WB = 5: table
# [P .. 16P], 26 positions per GLV half, 5 doublings per position; make variant NAME=k256w5 TU=ops_k256 DEFS="-DK256_WB=5") against 4-bit windows.
#   gpurun --timeout 900 -- 'mkdir -p gpurun_out/r4 && bash tools/ab_round4c.sh > gpurun_out/r4/ab_4.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
B="--no-cpu-baseline --no-other-configs --no-host-io --steps 5 --warmup 1"
line() { python -c 'import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("   ", d["config"]["workload"][:40], "ms_per_step", round(d["ms_per_step"], 3), "kernel_ms", round(d["roofline"]["kernel_ms"], 3), "value", "%.4g" % d["value"], "parity", d["parity_ok"])'; }
echo "#### correctness of the variant: the secp256k1 GPU tests against libecgpu_k256w5.so"
ECGPU_LIB=$PK/lib_exp/libecgpu_k256w5.so timeout -k 10 400 python -m pytest tests/test_gpu_k256.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do
  for v in default k256w5; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    echo "#### k256 variable base 2^24 (headline), library: $v (pass $rep)"
    timeout -k 10 200 python bench.py $B --workload k256_varbase 2>/dev/null | line
  done
done
for v in default k256w5; do
  if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
  echo "#### two-term linear combination 2^22, verification 2^22, library: $v"
  timeout -k 10 120 python tools/gpu_quick.py k256 22 lincomb2 2>&1 | tail -1
  timeout -k 10 120 python tools/gpu_quick.py k256 22 ecdsa 2>&1 | grep verify | tail -1
done
