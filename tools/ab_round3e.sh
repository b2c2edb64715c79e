#!/bin/bash
# Round-3 A/B measurements, fifth set (after the co-Z tables took the general addition out of the NIST variable-base kernels):
#   p384vb3   -DVB_WAVES=3: public-data P-384 variable base at 3 waves per SIMD (168 VGPRs)
#   p384ct4   "-DVBCT_WAVES(C)=4": constant-time P-384 variable base at 4 waves per SIMD (128 VGPRs) instead of 3
#   p256sqr   -DECGPU_P256_DEDICATED_SQR: P-256 squaring as 36 products + a reduction-only pass (P-384's form) instead of the general multiplication
#   gpurun --timeout 900 -- 'mkdir -p gpurun_out/r3 && bash tools/ab_round3e.sh > gpurun_out/r3/ab_e2.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
B="--no-cpu-baseline --no-other-configs --steps 5 --warmup 1"
line() { python -c 'import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("   ", d["config"]["workload"][:40], "ms_per_step", round(d["ms_per_step"], 3), "value", "%.4g" % d["value"], "parity", d["parity_ok"])'; }
for rep in 1 2; do
  for v in default p384vb3; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    echo "#### p384 variable base 2^22 (config 5), library: $v (pass $rep)"
    timeout -k 10 200 python bench.py $B --workload p384_varbase 2>/dev/null | line
  done
done
for v in default p384ct4; do
  if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
  echo "#### P-384 secret-scalar variable base 2^21, library: $v"
  timeout -k 10 300 python tools/ct_varbase_bench.py 21 p384 2>&1 | grep -v amdgpu.ids
done
for rep in 1 2; do
  for v in default p256sqr; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    echo "#### P-256, library: $v (pass $rep)"
    timeout -k 10 200 python bench.py $B --workload p256_fixedbase 2>/dev/null | line
    timeout -k 10 300 python tools/ct_varbase_bench.py 22 p256 2>&1 | grep -v amdgpu.ids
  done
done
