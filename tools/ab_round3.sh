#!/bin/bash
# Round-3 A/B measurements on one box: every variant is a separately linked library selected through
# ECGPU_LIB (the Python binding's switch for exactly this), same process order, same inputs.
#   default                  digits of the NIST variable-base kernels in LDS (DigitMem), k256 table with beta*x slots (1 KB per lane)
#   lib_exp/libecgpu_regdig  -DECGPU_DIGITS_IN_REGISTERS: the round-2 form (NW VGPRs + a select chain per read)
#   lib_exp/libecgpu_nobeta  -DECGPU_K256_NO_BETA_SLOTS: 512-byte k256 table, beta*x multiplied in on the lambda half
#   gpurun --timeout 900 -- 'bash tools/ab_round3.sh > gpurun_out/r3/ab.txt 2>&1'
# Build the variants first, in the container (they travel with the snapshot):  bash tools/ab_round3.sh build
cd "${GRAFT_REPO_ROOT:-.}"
PK=rustcrypto-elliptic-curves_amd
if [ "$1" = build ]; then
  cd $PK && mkdir -p build_exp lib_exp
  F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off"
  /opt/rocm/bin/hipcc $F -DECGPU_DIGITS_IN_REGISTERS -c csrc/ops_p256.hip -o build_exp/ops_p256_regdig.o &
  /opt/rocm/bin/hipcc $F -DECGPU_DIGITS_IN_REGISTERS -c csrc/ops_p384.hip -o build_exp/ops_p384_regdig.o &
  /opt/rocm/bin/hipcc $F -DECGPU_K256_NO_BETA_SLOTS -c csrc/ops_k256.hip -o build_exp/ops_k256_nobeta.o &
  wait
  L="-shared -fPIC --offload-arch=gfx950 -Wl,--version-script=csrc/ecgpu.map"
  /opt/rocm/bin/hipcc $L build/ecgpu.o build/ops_k256.o build_exp/ops_p256_regdig.o build_exp/ops_p384_regdig.o build/msm_k256.o build/msm_p256.o build/msm_p384.o -o lib_exp/libecgpu_regdig.so
  /opt/rocm/bin/hipcc $L build/ecgpu.o build_exp/ops_k256_nobeta.o build/ops_p256.o build/ops_p384.o build/msm_k256.o build/msm_p256.o build/msm_p384.o -o lib_exp/libecgpu_nobeta.so
  exit 0
fi
for rep in 1 2; do
  for v in default regdig; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PWD/$PK/lib_exp/libecgpu_$v.so; fi
    echo "== NIST variable base, library: $v (pass $rep)"
    timeout -k 10 200 python tools/ct_varbase_bench.py 22 p256,p384 2>&1 | grep -v amdgpu.ids
  done
done
for rep in 1 2; do
  for v in default nobeta; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PWD/$PK/lib_exp/libecgpu_$v.so; fi
    echo "== k256 headline, library: $v (pass $rep)"
    timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print('   ms_per_step %.3f  kernel_ms %.3f  value %.4e  parity_ok %s' % (j['ms_per_step'], r['kernel_ms'], j['value'], j['parity_ok']))"
  done
done
