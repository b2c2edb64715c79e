#!/bin/bash
# Round-3 A/B measurements, third set: the co-Z table chain of the secp256k1 variable-base kernels (public-data and constant-time)
# and XYZZ accumulators in the fixed-base kernels (wide tables: config 3; constant-time: signing).  The "old" libraries link the
# translation unit of the previous commit (built from a worktree of HEAD~) against the current objects, selected through ECGPU_LIB.
#   gpurun --timeout 900 -- 'mkdir -p gpurun_out/r3 && bash tools/ab_round3c.sh > gpurun_out/r3/ab_c.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
B="--no-cpu-baseline --no-other-configs --steps 5 --warmup 1"
line() { python -c 'import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("   ", d["config"]["workload"][:40], "ms_per_step", round(d["ms_per_step"], 3), "value", "%.4g" % d["value"], "parity", d["parity_ok"])'; }
for rep in 1 2; do
  for v in default oldk256; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    echo "#### k256 variable base 2^24 (headline), library: $v (pass $rep)"
    timeout -k 10 200 python bench.py $B --workload k256_varbase 2>/dev/null | line
  done
  for v in default oldp256; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    echo "#### p256 fixed base 2^24 (config 3), library: $v (pass $rep)"
    timeout -k 10 200 python bench.py $B --workload p256_fixedbase 2>/dev/null | line
  done
done
for v in default oldk256; do
  if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
  echo "#### k256 secret-scalar variable base 2^22, library: $v"
  timeout -k 10 200 python tools/ct_varbase_bench.py 22 k256 2>&1 | grep -v amdgpu.ids
done
for v in default oldk256 oldp256; do
  if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
  echo "#### signing / fixed base at 2^20 (oldk256: k256 rows are the previous kernels; oldp256: p256 rows), library: $v"
  timeout -k 10 300 python tools/util_bench.py 20 2>&1 | grep -E "ecdsa sign|mul_by_generator \(throughput"
done
