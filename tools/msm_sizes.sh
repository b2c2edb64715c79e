#!/bin/bash
# MSM time against the number of terms for both window widths (where 19-bit windows start to pay) and the number of
# bucket-sum runs per lane.   gpurun --timeout 900 -- 'bash tools/msm_sizes.sh'
cd "${GRAFT_REPO_ROOT:-.}"
for r in ${ROUNDS:-4 6 8}; do
  echo "== k256 2^23, ECGPU_MSM_ROUNDS=$r"
  ECGPU_MSM_CBITS=19 ECGPU_MSM_ROUNDS=$r timeout -k 10 120 python tools/gpu_quick.py k256 23 msm 2>&1 | grep msm | tail -2
done
for cv in ${CURVES:-k256 p256 p384}; do
  for lg in ${SIZES:-18 19 20 21 22 23 24}; do
    if [ "$cv" = p384 ] && [ "$lg" -gt 22 ]; then continue; fi
    for cb in 16 19; do
      echo "== $cv 2^$lg ECGPU_MSM_CBITS=$cb"
      ECGPU_MSM_SMALL=0 ECGPU_MSM_CBITS=$cb timeout -k 10 200 python tools/gpu_quick.py $cv $lg msm 2>&1 | grep msm | tail -1
    done
  done
done
