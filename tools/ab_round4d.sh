#!/bin/bash
# Round-4 A/B measurements, set five: phase stagger of the headline kernel's workgroups (kernels.hpp K256_STAGGER: the four workgroups of a CU start
# 0 / 1 / 2 / 3 x K256_STAGGER x 3.4 us apart), prompted by two contexts on one card running 2 x 2^24 units in 239.6 ms = 119.8 ms per 2^24
# (bench.py --gpus 2 --inproc --same-device) where one context alone needs 130.8 ms.
cd "${GRAFT_REPO_ROOT:-.}"
PK=$PWD/rustcrypto-elliptic-curves_amd
B="--no-cpu-baseline --no-other-configs --no-host-io --steps 5 --warmup 1"
line() { python -c 'import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("   ", d["config"]["workload"][:40], "ms_per_step", round(d["ms_per_step"], 3), "kernel_ms", round(d["roofline"]["kernel_ms"], 3), "value", "%.4g" % d["value"], "parity", d["parity_ok"])'; }
for rep in 1 2 3; do
  for v in default $VARIANTS; do
    if [ $v = default ]; then unset ECGPU_LIB; else export ECGPU_LIB=$PK/lib_exp/libecgpu_$v.so; fi
    echo "#### k256 variable base 2^24 (headline), library: $v (pass $rep)"
    timeout -k 10 200 python bench.py $B --workload k256_varbase 2>/dev/null | line
  done
done
