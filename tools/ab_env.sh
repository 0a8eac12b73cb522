#!/bin/bash
# A/B of environment switches of ONE library build in one GPU call: tools/ab_env.sh "VAR=1 VAR2=x" "VAR=0" ...   (each: lone frames +
# 8-frame calls of tools/gpu_ab.py, the bench's timed region, one rank's eighth of a 20-frame burst); twice round, so that drift shows
for round in 1 2; do
  for envs in "$@"; do
    echo "== [$envs] (round $round)"
    env $envs timeout -k 10 200 python tools/gpu_ab.py "$envs" 2>&1 | grep -v amdgpu.ids || exit 1
    env $envs timeout -k 10 200 python bench.py --no-config4 --no-other-configs --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value'],1), 'Mrays/s', d['config']['ms_per_step_samples'], 'lone', round(d['config']['single_frame_latency_ms'],4), 'checksum ok', d['config']['frame_checksum_ok'], 'quad ms/call', round(d['roofline']['kernels'][0]['ms_per_call'],3))" || exit 1
    env $envs RANKSHARE_F=2 RANKSHARE_B=10 RANKSHARE_K=20 RANKSHARE_ROTATE=1 RANKSHARE_REPEAT=21 timeout -k 10 120 python tools/gpu_rank_share.py 8 2>&1 | grep parts || exit 1
  done
done
