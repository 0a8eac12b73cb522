#!/bin/bash
# per-component VALU instruction counts of k_trace_easy: one rocprofv3 --pmc run per scene variant (tools/prof_easy_components.py)
# usage: bash tools/prof_easy_components.sh <tag>   -> gpurun_out/<tag>/, gpurun_out/<tag>.md
set -u
export TMPDIR=/tmp
out=gpurun_out/$1
mkdir -p $out
for v in v0_raygen_black v1_sky v2_plane_uniform_nolight v3_plane_checker_nolight v4_plane_checker_sun v5_plus_bunny_level0 v6_config3; do
  timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $out/$v -- python3 tools/prof_easy_components.py run $v > $out/$v.log 2>&1 || { echo "variant $v failed"; tail -3 $out/$v.log; exit 1; }
done
python3 tools/prof_easy_components.py summary $out > gpurun_out/$1.md
cat gpurun_out/$1.md
