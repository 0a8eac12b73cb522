#!/bin/bash
# the bench's timed region against frames per call x calls in flight, for the driver's --steps 20 and for --steps 200
for steps in 20 200; do
  for b in 4 5 7 8 10 16 20 32; do
    for f in 2 3 4; do
      v=$(timeout -k 10 120 python bench.py --steps $steps --warmup 20 --batch $b --inflight $f --no-extras --no-cpu-baseline --repeats 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4))") || exit 1
      echo "steps $steps batch $b inflight $f: $v"
    done
  done
done
