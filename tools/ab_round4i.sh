#!/bin/bash
# Round-4 A/B measurements, set eleven: two switches of the headline kernel that had not been measured since the dynamic work distribution / the 5-bit windows.
#   waves3   = ECGPU_OPT_K256_WAVES 3 (run-time option: k256_mul_fast_kernel<32,3>, 168 VGPRs) - round 3 found 3 and 4 waves per SIMD equal, but that was with
#              the static grid stride whose favoured waves left early (profiles/r04_ab_measurements.txt set six); with every wave busy to the end it is a new question
#   nobeta   = -DECGPU_K256_NO_BETA_SLOTS (make variant NAME=nobeta TU=ops_k256 DEFS="-DECGPU_K256_NO_BETA_SLOTS"): 16 table slots of (x, y) instead of 32 with
#              beta x beside x: 1 KB instead of 2 KB of table per unit, 16 multiplications less in the table, one more on every addition of the lambda half
#              (measured 1.2 % slower with 8 entries in round 3, never with 16)
#   gpurun --timeout 900 -- 'mkdir -p gpurun_out/r4 && bash tools/ab_round4i.sh > gpurun_out/r4/ab_11.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
echo "#### correctness of the variant: the secp256k1 GPU tests against libecgpu_nobeta.so"
ECGPU_LIB=$PK/lib_exp/libecgpu_nobeta.so timeout -k 10 400 python -m pytest tests/test_gpu_k256.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do
  for v in default waves3 nobeta; do
    unset ECGPU_LIB ECGPU_K256_FAST_WAVES
    [ $v = nobeta ] && export ECGPU_LIB=$PK/lib_exp/libecgpu_nobeta.so
    [ $v = waves3 ] && export ECGPU_K256_FAST_WAVES=3
    echo "#### k256 variable base 2^24 (headline), $v (pass $rep)"
    timeout -k 10 200 python tools/gpu_quick.py k256 24 var 2>&1 | grep "n=2" | tail -2
  done
done
