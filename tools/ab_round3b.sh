#!/bin/bash
# Round-3 A/B measurements, second set: what the 64-byte gathers of the MSM bucket sums cost and why (the clock), occupancy
# against spills, non-temporal gathers.  Variants are separately linked libraries selected through ECGPU_LIB.
#   bash tools/ab_round3b.sh build          (in the container; the libraries travel with the snapshot)
#   gpurun --timeout 900 -- 'mkdir -p gpurun_out/r3 && bash tools/ab_round3b.sh > gpurun_out/r3/ab_b.txt 2>&1'
#     w3          -DMSM_BS_WAVES=3: bucket sums at 3 waves per SIMD (150 VGPRs, no spill) instead of 4 (128 VGPRs + 110 spilled)
#     hot<mask>   -DMSM_HOT_GATHER=<mask>: DIAGNOSTIC, wrong sums - every gather's term index masked, i.e. the gathers served from
#                 a footprint of (mask + 1) x 64 B per half: 0x3FF 64 KB (L1/L2), 0x7FFF 2 MB (L2), 0xFFFFF 64 MB (Infinity Cache),
#                 0x3FFFFF 256 MB; the real footprint at 2^23 terms is 512 MB per half
#     p384w2      -DVB_WAVES=2 -DVBCT_WAVES(C)=2: the P-384 variable-base kernels at 2 waves per SIMD (256 VGPRs, ~30 spilled)
cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
MASKS="0x3FF 0x7FFF 0xFFFFF 0x3FFFFF"
if [ "$1" = build ]; then
  cd $PK && mkdir -p build_exp lib_exp
  F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off"
  /opt/rocm/bin/hipcc $F -DMSM_BS_WAVES=3 -c csrc/msm_k256.hip -o build_exp/msm_k256_w3.o &
  for m in $MASKS; do /opt/rocm/bin/hipcc $F -DMSM_HOT_GATHER=$m -c csrc/msm_k256.hip -o build_exp/msm_k256_hot$m.o & done
  /opt/rocm/bin/hipcc $F -DVB_WAVES=2 "-DVBCT_WAVES(C)=2" -c csrc/ops_p384.hip -o build_exp/ops_p384_w2.o &
  wait
  L="-shared -fPIC --offload-arch=gfx950 -Wl,--version-script=csrc/ecgpu.map"
  O="build/ecgpu.o build/ops_k256.o build/ops_p256.o"
  /opt/rocm/bin/hipcc $L $O build/ops_p384.o build_exp/msm_k256_w3.o build/msm_p256.o build/msm_p384.o -o lib_exp/libecgpu_w3.so
  for m in $MASKS; do /opt/rocm/bin/hipcc $L $O build/ops_p384.o build_exp/msm_k256_hot$m.o build/msm_p256.o build/msm_p384.o -o lib_exp/libecgpu_hot$m.so; done
  /opt/rocm/bin/hipcc $L $O build_exp/ops_p384_w2.o build/msm_k256.o build/msm_p256.o build/msm_p384.o -o lib_exp/libecgpu_p384w2.so
  exit 0
fi
for v in default w3 $(for m in $MASKS; do echo hot$m; done); do
  if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
  echo "#### k256 MSM 2^23 terms, library: $v   (hot*: parity_ok False is expected)"
  bash tools/clock_probe.sh "k256_msm:1200"
done
for rep in 1 2; do
  for v in default p384w2; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    echo "#### P-384 variable base, library: $v (pass $rep)"
    timeout -k 10 200 python tools/ct_varbase_bench.py 22 p384 2>&1 | grep -v amdgpu.ids
  done
done
