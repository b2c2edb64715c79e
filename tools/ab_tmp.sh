cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
timeout -k 10 600 python -m pytest tests/test_gpu_ecdsa.py tests/test_gpu_api.py tests/test_gpu_k256.py tests/test_gpu_nist.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
for v in default fbw; do
  if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
  echo "== library $v (pass $rep)"
  timeout -k 10 300 python tools/util_bench.py 20 2>&1 | grep "ecdsa sign\|mul_by_generator"
done
done
