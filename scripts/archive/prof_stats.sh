#!/bin/bash
# usage: prof_stats.sh <tag> [bench args]  -- rocprofv3 kernel-trace stats of the default bench
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stats_$tag -o $tag -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/stats_$tag.log 2>&1
grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*\|"pipeline_ms": [0-9.]*\|"ms_per_step": [0-9.]*' $R/gpurun_out/stats_$tag.log | tr '\n' ' '; echo
python $R/scripts/rocpd_stats.py $R/gpurun_out/stats_$tag/${tag}_results.db | grep -E "prad|rocclr|kernel \|" | head -20
