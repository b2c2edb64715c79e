#!/bin/bash
# Round-3 A/B measurements, fourth set: co-Z table chains with one inverted denominator per table in the P-256 / P-384
# variable-base kernels (public-data and constant-time).  "old" libraries: the curve's translation unit of commit 703598f
# linked against the current objects, selected through ECGPU_LIB.
#   gpurun --timeout 900 -- 'mkdir -p gpurun_out/r3 && bash tools/ab_round3d.sh > gpurun_out/r3/ab_d.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
B="--no-cpu-baseline --no-other-configs --steps 5 --warmup 1"
line() { python -c 'import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("   ", d["config"]["workload"][:40], "ms_per_step", round(d["ms_per_step"], 3), "value", "%.4g" % d["value"], "parity", d["parity_ok"])'; }
for rep in 1 2; do
  for v in default oldp384; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    echo "#### p384 variable base 2^22 (config 5), library: $v (pass $rep)"
    timeout -k 10 200 python bench.py $B --workload p384_varbase 2>/dev/null | line
  done
done
for v in default oldp384 oldp256; do
  if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
  echo "#### secret-scalar / reference / public variable base, P-256 at 2^22 and P-384 at 2^21 (oldp256: the P-256 rows are the previous kernels; oldp384: the P-384 rows), library: $v"
  timeout -k 10 300 python tools/ct_varbase_bench.py 22 p256 2>&1 | grep -v amdgpu.ids
  timeout -k 10 300 python tools/ct_varbase_bench.py 21 p384 2>&1 | grep -v amdgpu.ids
done
for v in default oldp384; do
  if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
  echo "#### P-384 signing / fixed base at 2^20 (XYZZ accumulators), library: $v"
  timeout -k 10 300 python tools/util_bench.py 20 2>&1 | grep -E "^p384 .*(ecdsa sign|ecdsa verify|mul_by_generator \(throughput)"
done
