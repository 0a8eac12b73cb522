#!/bin/bash
# HBM traffic of bench.py's kernels from PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc runs (never combined with tracing), FETCH_SIZE doubled on gfx950.
# usage (on the GPU box): bash tools/prof_traffic.sh <tag>   -> gpurun_out/<tag>/{fetch,write}/... + traffic.json
set -u
export TMPDIR=/tmp
tag=${1:-traffic}
out=gpurun_out/$tag
mkdir -p $out
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $out/fetch -- python3 bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-extras > $out/fetch.log 2>&1 || { echo "fetch pass failed"; tail -3 $out/fetch.log; exit 1; }
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum --output-format csv -d $out/write -- python3 bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-extras > $out/write.log 2>&1 || { echo "write pass failed"; tail -3 $out/write.log; exit 1; }
python3 tools/prof_traffic.py $out
