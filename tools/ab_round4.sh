#!/bin/bash
# Round-4 A/B measurements, set two, alternating runs on one box:
#   p256 fixed base 2^24 (config 3): gather prefetch and / or 3 waves per SIMD against the shipped kernel
#     fbpf3 = -DECGPU_FB_PREFETCH -DFB_WIDE_WAVES=3, fbpf4 = -DECGPU_FB_PREFETCH (4 waves), fbw3 = -DFB_WIDE_WAVES=3 (make variant ...)
#   k256 MSM 2^23: DIAGNOSTIC build msm1arr (-DMSM_DIAG_ONE_ARRAY: both GLV halves gather from one array of 512 MB; wrong sums, same
#     instruction stream): the most a half-aware single point array could buy
#   gpurun --timeout 900 -- 'mkdir -p gpurun_out/r4 && bash tools/ab_round4.sh > gpurun_out/r4/ab_2.txt 2>&1'
cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
B="--no-cpu-baseline --no-other-configs --no-host-io --steps 8 --warmup 2"
line() { python -c 'import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("   ", d["config"]["workload"][:40], "ms_per_step", round(d["ms_per_step"], 3), "kernel_ms", round(d["roofline"]["kernel_ms"], 3), "value", "%.4g" % d["value"], "parity", d["parity_ok"])'; }
for rep in 1 2 3; do
  for v in default fbpf3 fbpf4 fbw3; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    echo "#### p256 fixed base 2^24 (config 3), library: $v (pass $rep)"
    timeout -k 10 200 python bench.py $B --workload p256_fixedbase 2>/dev/null | line
  done
done
for rep in 1 2 3; do
  for v in default msm1arr; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    echo "#### k256 MSM 2^23 terms, library: $v (pass $rep)"
    timeout -k 10 120 python tools/gpu_quick.py k256 23 msm 2>&1 | grep "msm:" | tail -2
  done
done
