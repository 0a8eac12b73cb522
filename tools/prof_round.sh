#!/bin/bash
# One round's measurements on the GPU box: the default bench line, the rocprofv3 kernel-trace stats of the same timed region
# (`bench.py --no-extras`: only launches of the timed region's size), and the PMC passes (one run per counter group) of the
# bench's shape of work (config 3, 32-frame calls; FETCH_SIZE and WRITE_SIZE in passes of their own), of lone config-4 frames.
# (config 4 with the device-built BVH, the library's default for a mesh of that size and the one bench.py reports)
# usage: bash tools/prof_round.sh <tag>      -> gpurun_out/<tag>_*
set -u
export TMPDIR=/tmp
tag=${1:-r03}
o=gpurun_out
timeout -k 10 300 python3 bench.py > $o/${tag}_bench.json 2> $o/${tag}_bench.err || { echo "bench failed"; tail -5 $o/${tag}_bench.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_stats -- python3 bench.py --steps 192 --no-extras --no-cpu-baseline > $o/${tag}_stats.json 2> $o/${tag}_stats.err || { echo "stats run failed"; tail -5 $o/${tag}_stats.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_stats_inflight1 -- python3 bench.py --steps 192 --no-extras --no-cpu-baseline --inflight 1 > $o/${tag}_stats_inflight1.json 2> $o/${tag}_stats_inflight1.err || { echo "stats run (inflight 1) failed"; exit 1; }
bash tools/prof_passes.sh ${tag}_pmc_cfg3_b32 3 2 32 > $o/${tag}_pmc_cfg3_b32.txt 2>&1 || { echo "pmc cfg3 failed"; tail -5 $o/${tag}_pmc_cfg3_b32.txt; exit 1; }
python3 tools/prof_summary.py $o/${tag}_pmc_cfg3_b32 64 $o/${tag}_pmc_summary_config3_batch32.json > $o/${tag}_pmc_summary_config3_batch32.md
bash tools/prof_passes.sh ${tag}_pmc_cfg4 4 3 > $o/${tag}_pmc_cfg4.txt 2>&1 || { echo "pmc cfg4 failed"; tail -5 $o/${tag}_pmc_cfg4.txt; exit 1; }
python3 tools/prof_summary.py $o/${tag}_pmc_cfg4 3 $o/${tag}_pmc_summary_config4.json > $o/${tag}_pmc_summary_config4.md
# config 5 (10.5 M triangles, 8K, one frame per pass: ~40 s of scene construction each)
if [ "${PROF_CFG5:-1}" = "1" ]; then
  bash tools/prof_passes.sh ${tag}_pmc_cfg5 5 1 > $o/${tag}_pmc_cfg5.txt 2>&1 || { echo "pmc cfg5 failed"; tail -5 $o/${tag}_pmc_cfg5.txt; exit 1; }
  python3 tools/prof_summary.py $o/${tag}_pmc_cfg5 1 $o/${tag}_pmc_summary_config5.json > $o/${tag}_pmc_summary_config5.md
fi
echo "round profile done"
