#!/bin/bash
# per-kernel device times of whole 256^3 cases (Original + 8 wavelet sub-bands, six classes): class-after-class route (launches
# do not overlap: durations are kernel speeds) and the case pipeline (three side streams: launches overlap)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04case; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
{
echo "# r04 -- rocprofv3 --kernel-trace --stats of scripts/case_latency.py (10 cases each; scripts/prof_case_r04.sh)"
echo
for mode in 0 1; do
  rm -rf /tmp/cl
  CASE_ONLY=$mode CASE_REP=9 rocprofv3 --kernel-trace --stats -d /tmp/cl -o s -- python $R/scripts/case_latency.py > /tmp/cl.log 2>&1
  if [ $mode = 0 ]; then echo "## class after class (enqueueSegment: False); totals are per 10 cases"; else echo "## case pipeline (default); kernels of different streams overlap, durations are not kernel speeds"; fi
  grep "enqueueSegment=" /tmp/cl.log | sed 's/^/(under the profiler) /'
  echo
  python $R/scripts/rocpd_stats.py /tmp/cl/s_results.db | head -45
  echo
done
} > $O/case_kernels.md
cd $R; python scripts/case_latency.py 2>&1 | grep enqueueSegment= >> $O/case_kernels.md
