#!/bin/bash
# the driver's regime (bench.py --steps 20 --warmup 5) under environment switches / call shapes:
#   tools/sweep_k20.sh "ENV=.. --batch 10" "ENV=.. --batch 20 --inflight 1" ...   (words with '=' are environment, the rest bench.py flags)
for spec in "$@"; do
  envs=""; flags=""
  for w in $spec; do case "$w" in *=*) envs="$envs $w";; *) flags="$flags $w";; esac; done
  echo "== [$spec]"
  for rep in 1 2; do
    env $envs timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline $flags 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  bench K=20', round(d['value'],1), 'Mrays/s', [round(x,4) for x in d['config']['ms_per_step_samples']], 'B', d['config']['frames_per_call'], 'F', d['config']['calls_in_flight'])" || exit 1
  done
done
