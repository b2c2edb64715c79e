#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
# Round-3 A/B measurements, sixth set (second half): the fused difference of products in the XYZZ additions (MSM bucket sums, signing) and in the
# constant-time secp256k1 variable-base kernel.  "fused" = the library of tools/ab_round3f.sh (only the Jacobian mixed addition of the headline
# kernel fused), default = the in-tree build.   gpurun --timeout 900 -- 'bash tools/ab_round3g.sh > gpurun_out/r3/ab_g.txt 2>&1'
PK=$PWD/rustcrypto-elliptic-curves_amd
B="--no-cpu-baseline --no-other-configs --steps 10 --warmup 2"
line() { python -c 'import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("   ", d["config"]["workload"][:40], "ms_per_step", round(d["ms_per_step"], 3), "value", "%.4g" % d["value"], "parity", d["parity_ok"])'; }
for rep in 1 2; do
  for v in default fused; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    echo "#### k256 MSM 2^23, library: $v (pass $rep)  [fused = only the Jacobian mixed addition of the headline kernel fused; default = XYZZ / CT additions too]"
    timeout -k 10 200 python bench.py $B --workload k256_msm 2>/dev/null | line
  done
done
for v in default fused; do
  if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
  echo "#### k256 CT variable base 2^22; signing 2^20; library: $v"
  timeout -k 10 200 python tools/ct_varbase_bench.py 22 k256 2>&1 | grep -v amdgpu.ids | head -3
  timeout -k 10 300 python tools/util_bench.py 20 2>&1 | grep -E "^k256 .*(ecdsa sign|ecdsa verify|mul_by_generator \(throughput)"
done
