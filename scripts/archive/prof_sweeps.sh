#!/bin/bash
# usage: prof_sweeps.sh <tag> [bench args] -- per-dispatch sweep kernel durations of one bench run
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sw_$tag -o $tag -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/sw_$tag.log 2>&1
grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*\|"pipeline_ms": [0-9.]*' $R/gpurun_out/sw_$tag.log | tr '\n' ' '; echo
python $R/scripts/rocpd_stats.py $R/gpurun_out/sw_$tag/${tag}_results.db | grep -E "prad" | head -8
python $R/scripts/rocpd_dispatches.py $R/gpurun_out/sw_$tag/${tag}_results.db sweep_ 16
