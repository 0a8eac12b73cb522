#!/bin/bash
# like tools/ab_env.sh without the rank-share burst, one round: tools/ab_env_quick.sh "VAR=1" "VAR=0" ...
for envs in "$@"; do
  echo "== [$envs]"
  env $envs timeout -k 10 200 python tools/gpu_ab.py "$envs" 2>&1 | grep -v amdgpu.ids || exit 1
  env $envs timeout -k 10 200 python bench.py --no-config4 --no-other-configs --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value'],1), 'Mrays/s', [round(x,4) for x in d['config']['ms_per_step_samples']], 'lone', round(d['config']['single_frame_latency_ms'],4), 'checksum ok', d['config']['frame_checksum_ok'], 'quad ms/call', round(d['roofline']['kernels'][0]['ms_per_call'],3))" || exit 1
done
