# A/B set ten: cooperative gather of the fixed-base table entries (four lanes fetch one 64-byte entry with one 16-byte load each, LDS transpose): -DECGPU_FB_COOP_GATHER
cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
B="--no-cpu-baseline --no-other-configs --no-host-io --steps 6 --warmup 1"
for rep in 1 2 3; do
  for v in default fbcoop; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    timeout -k 10 200 python bench.py $B --workload p256_fixedbase 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('p256 fixed 2^24', '$v', 'pass $rep', round(d['ms_per_step'],3), 'ms', d['parity_ok'])"
  done
done
