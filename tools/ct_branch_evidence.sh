#!/bin/bash
# Branch-free secp256k1 field operations in the constant-time kernels: counters for four random scalar sets and
# the cost, shipped build (branch-free) against the A/B build that keeps the rare carry branches (make -C rustcrypto-elliptic-curves_amd ab-ctbranch).
#   gpurun --timeout 900 -- 'bash tools/ct_branch_evidence.sh > gpurun_out/r4/ct_branch.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
PK=$PWD/rustcrypto-elliptic-curves_amd
for v in branchfree ctbranch; do
  if [ $v = branchfree ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_ctbranch.so; fi
  echo "######## counters, library: $v ($( [ $v = branchfree ] && echo 'shipped: ECGPU_K256_BRANCHFREE' || echo 'A/B build: rare carry paths as branches' ))"
  rm -rf gpurun_out/ctb_$v
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/ctb_$v -- python3 tools/ct_branch_evidence.py > gpurun_out/ctb_$v.log 2>&1 || { echo "profile failed"; tail -5 gpurun_out/ctb_$v.log; }
  python3 tools/ct_summarize.py gpurun_out/ctb_$v
  rm -rf gpurun_out/ctb_$v
done
for rep in 1 2 3; do
  for v in branchfree ctbranch; do
    if [ $v = branchfree ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_ctbranch.so; fi
    echo "#### time, library: $v (pass $rep): ECDH kernel 2^22, signing 2^20"
    timeout -k 10 200 python tools/ct_varbase_bench.py 22 k256 2>&1 | grep -v amdgpu.ids | head -2
    timeout -k 10 120 python tools/gpu_quick.py k256 20 ecdsa 2>&1 | grep sign | tail -1
  done
done
