#!/bin/bash
# Round-4 A/B measurements, set twelve: the many-term (shared-doubling) kernels at 3 instead of 4 waves per SIMD.
#   straus3 = curve_ops.hpp::lincomb_straus with WAVES = 3 for all three curves (straus::lincomb_kernel<C, 3>: 168 VGPRs; secp256k1 spills 22 VGPRs
#             instead of 119, P-384 7, P-256 0) - the known-gaps note of DESIGN section 7 on the spills of the secp256k1 kernel
#   gpurun --timeout 900 -- 'mkdir -p gpurun_out/r4 && bash tools/ab_round4j.sh > gpurun_out/r4/ab_12.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
for rep in 1 2; do
  for v in default straus3; do
    unset ECGPU_LIB
    [ $v = straus3 ] && export ECGPU_LIB=$PK/lib_exp/libecgpu_straus3.so
    echo "#### linear combinations of 3 .. 1024 terms, 2^22 terms in all, $v (pass $rep)"
    timeout -k 10 300 python tools/lincomb_bench.py 22 k256,p256 2>&1 | grep lincomb | cut -c1-110
  done
done
