#!/bin/bash
# A/B of library builds in ONE GPU call: tools/ab_round.sh build_ab/a.so build_ab/b.so ...   (each: lone frames + 8-frame calls of
# tools/gpu_ab.py, the bench's timed region, one rank's eighth of a 20-frame burst); twice round, so that drift shows
for round in 1 2; do
  for so in "$@"; do
    echo "== $so (round $round)"
    RTHIP_SO=$so timeout -k 10 200 python tools/gpu_ab.py $so 2>&1 | grep -v amdgpu.ids || exit 1
    RTHIP_SO=$so timeout -k 10 200 python bench.py --no-config4 --no-other-configs --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value'],1), 'Mrays/s', d['config']['ms_per_step_samples'], 'lone', round(d['config']['single_frame_latency_ms'],4), 'checksum ok', d['config']['frame_checksum_ok'])" || exit 1
    RTHIP_SO=$so RANKSHARE_F=2 RANKSHARE_B=10 RANKSHARE_K=20 RANKSHARE_ROTATE=1 RANKSHARE_REPEAT=21 timeout -k 10 120 python tools/gpu_rank_share.py 8 2>&1 | grep parts || exit 1
  done
done
