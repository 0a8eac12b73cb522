#!/bin/bash
# the bench's timed region with the workspace chunk limit raised (RTHIP_CHUNK_RAYS), so that calls of more than 8 frames stay one chunk
export RTHIP_CHUNK_RAYS=268435456
for b in 48 64; do
  for f in 2 3; do
    v=$(timeout -k 10 120 python bench.py --steps 192 --warmup 64 --batch $b --inflight $f --no-extras --no-cpu-baseline --repeats 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4))") || exit 1
    echo "steps 192 batch $b inflight $f: $v"
  done
done
for b in 5 7 10 20; do
  for f in 1 2 3 4; do
    v=$(timeout -k 10 120 python bench.py --steps 20 --warmup 20 --batch $b --inflight $f --no-extras --no-cpu-baseline --repeats 7 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4))") || exit 1
    echo "steps 20 batch $b inflight $f: $v"
  done
done
