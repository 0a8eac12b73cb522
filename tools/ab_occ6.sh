#!/bin/bash
# k_trace_quad at 5 / 6 waves per SIMD (RTHIP_QUAD_OCC6=0/1) on the scenes with millions of triangles (configs 4, 5) and the bunny (3)
for round in 1 2; do
  for v in 0 1; do
    echo "== RTHIP_QUAD_OCC6=$v (round $round)"
    RTHIP_QUAD_OCC6=$v AB_CONFIGS=4,3 timeout -k 10 200 python tools/gpu_ab.py occ6=$v 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
for v in 0 1; do
  RTHIP_QUAD_OCC6=$v AB_CONFIGS=5 timeout -k 10 300 python tools/gpu_ab.py occ6=$v 2>&1 | grep -v amdgpu.ids || exit 1
done
