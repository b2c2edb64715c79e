cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
B="--no-cpu-baseline --no-other-configs --no-host-io --steps 4 --warmup 1"
for rep in 1 2; do
  for v in default p384c1 p384c2 p384c3; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    timeout -k 10 200 python bench.py $B --workload p384_varbase 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('p384 2^22', '$v', 'pass $rep', round(d['ms_per_step'],3), 'ms', d['parity_ok'])"
  done
  for v in default p256c2; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    timeout -k 10 100 python tools/gpu_quick.py p256 22 var 2>&1 | grep "default:" | tail -2 | sed "s/^/   $v /"
  done
done
