#!/bin/bash
# bounded grids of the queue kernels (RTHIP_GRID_LEVEL: k_trace_easy / k_shade, RTHIP_GRID_QUAD: k_trace_quad) against the bench, a lone frame and config 4
for gl in 2048 4096 8192; do
  for gq in 4096 8192 16384; do
    v=$(RTHIP_GRID_LEVEL=$gl RTHIP_GRID_QUAD=$gq timeout -k 10 120 python bench.py --steps 192 --no-config4 --no-other-configs --no-cpu-baseline --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'lone', round(d['config']['single_frame_latency_ms'],4))") || exit 1
    w=$(RTHIP_GRID_LEVEL=$gl RTHIP_GRID_QUAD=$gq AB_CONFIGS=4 timeout -k 10 120 python tools/gpu_ab.py x 2>/dev/null | cut -d'|' -f2 | cut -c1-30)
    echo "level $gl quad $gq: bench $v |$w"
  done
done
