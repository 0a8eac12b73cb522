for b in 23 25 29 32 34 40 50 64; do
  v=$(timeout -k 10 120 python bench.py --steps 200 --batch $b --no-extras --no-cpu-baseline --repeats 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4))") || exit 1
  echo "steps 200 batch $b inflight 3: $v"
done
