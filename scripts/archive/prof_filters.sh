#!/bin/bash
# usage: prof_filters.sh <tag> -- rocprofv3 kernel trace of config 3 (scripts/bench_filters.py), per-dispatch LoG passes
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/filt_$tag -o $tag -- python $R/scripts/bench_filters.py --size 256 > $R/gpurun_out/filt_$tag.log 2>&1
tail -1 $R/gpurun_out/filt_$tag.log
python $R/scripts/rocpd_stats.py $R/gpurun_out/filt_$tag/${tag}_results.db | grep prad | head
python $R/scripts/rocpd_dispatches.py $R/gpurun_out/filt_$tag/${tag}_results.db rgauss 9
python $R/scripts/rocpd_dispatches.py $R/gpurun_out/filt_$tag/${tag}_results.db swt_axis 3
