#!/bin/bash
# Shader clock and socket power WHILE a bench workload runs (the kernels are power-limited: the sustained clock, not the
# nominal 2.4 GHz, is what a cycle count has to be priced at).  One line per sample; the workload's own result last.
#   gpurun --timeout 600 -- 'mkdir -p gpurun_out/r3 && bash tools/clock_probe.sh > gpurun_out/r3/clocks.txt 2>&1'
#   bash tools/clock_probe.sh "k256_msm:1500 k256_varbase:100"      (workload:steps; ECGPU_LIB selects a variant library)
cd "${GRAFT_REPO_ROOT:-.}"
LIST=${1:-"k256_varbase:110 p256_fixedbase:1100 k256_msm:1500 p384_varbase:60"}
for item in $LIST; do
  w=${item%%:*}; steps=${item##*:}
  echo "== $w ($steps steps)"
  timeout -k 10 200 python bench.py --workload $w --steps $steps --warmup 3 --no-cpu-baseline --no-other-configs > /tmp/clock_probe.json 2>/dev/null &
  BP=$!
  sleep 10
  for i in 1 2 3 4 5; do
    rocm-smi --showclocks --showpower 2>/dev/null | python3 -c "
import re,sys
t=sys.stdin.read()
s=re.search(r'sclk clock level: \d+: \((\d+)Mhz\)',t); m=re.search(r'mclk clock level: \d+: \((\d+)Mhz\)',t); p=re.search(r'Power \(W\): ([\d.]+)',t)
print('   sclk %s MHz  mclk %s MHz  socket power %s W' % (s and s.group(1), m and m.group(1), p and p.group(1)))"
    sleep 1
  done
  wait $BP
  python3 -c "
import json
j=json.loads(open('/tmp/clock_probe.json').read()); print('   ms_per_step %.3f  value %.4e %s  parity_ok %s' % (j['ms_per_step'], j['value'], j['unit'], j['parity_ok']))"
done
