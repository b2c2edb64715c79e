#!/bin/bash
# Round-3 A/B measurements, seventh set: Y3 of the secp256k1 doubling as one fused sum of two products, E (D - X3) + B (-8B)
# (the squaring C = B^2 becomes the second product of k256::mul_add2; now the default), against the squaring, the multiplication
# and the subtraction (-DECGPU_K256_NO_FUSED_DBL).  At the time of the measurement "fdbl" was ops_k256 built with the fused form and
# "default" the in-tree build without it, both linked against the same other objects.
#   gpurun --timeout 900 -- 'mkdir -p gpurun_out/r3 && bash tools/ab_round3h.sh > gpurun_out/r3/ab_h.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
B="--no-cpu-baseline --no-other-configs --steps 5 --warmup 1"
line() { python -c 'import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("   ", d["config"]["workload"][:40], "ms_per_step", round(d["ms_per_step"], 3), "value", "%.4g" % d["value"], "parity", d["parity_ok"])'; }
for rep in 1 2 3; do
  for v in default fdbl; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    echo "#### k256 variable base 2^24 (headline), library: $v (pass $rep)"
    timeout -k 10 200 python bench.py $B --workload k256_varbase 2>/dev/null | line
  done
done
for v in default fdbl; do
  if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
  echo "#### k256 secret-scalar variable base 2^22, two-term linear combination 2^22, verification 2^20, library: $v"
  timeout -k 10 200 python tools/ct_varbase_bench.py 22 k256 2>&1 | grep -v amdgpu.ids | head -3
  timeout -k 10 120 python tools/gpu_quick.py k256 22 lincomb2 2>&1 | tail -1
  timeout -k 10 120 python tools/gpu_quick.py k256 20 ecdsa 2>&1 | tail -1
done
