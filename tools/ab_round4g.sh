# A/B set nine: the constant-time fixed-base kernel (signing) launched with 1 / 2 / 4 workgroups per resident one
# (static work, fewer results per inversion against the better residency of an oversubscribed grid)
cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
for rep in 1 2; do
  for lg in 20 22; do
    for v in default k256fbct2 k256fbct4; do
      if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
      timeout -k 10 150 python tools/gpu_quick.py k256 $lg ecdsa 2>&1 | grep "ecdsa sign" | tail -1 | sed "s/^/$v pass $rep: /"
    done
    for v in default p256fbct2 p256fbct4; do
      if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
      timeout -k 10 150 python tools/gpu_quick.py p256 $lg ecdsa 2>&1 | grep "ecdsa sign" | tail -1 | sed "s/^/$v pass $rep: /"
    done
  done
done
