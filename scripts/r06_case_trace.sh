#!/bin/bash
# round 6: how busy is the GPU under one host thread of cases?  rocprofv3 kernel trace of scripts/r06_case_loop.py, union of
# the kernel intervals against the span (scripts/rocpd_busy.py), case after case and with one case of overlap
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06case; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for many in "" 1; do
  rm -rf /tmp/cl
  MANY=$many rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/cl -o s -- python $R/scripts/r06_case_loop.py 12 > /tmp/cl.log 2>&1
  echo "## MANY=$many"; grep "per case" /tmp/cl.log
  python $R/scripts/rocpd_busy.py /tmp/cl/s_results.db 0.3 | head -34
done > $O/busy.md 2>&1
cd $R
python scripts/r06_case_loop.py 12; MANY=1 python scripts/r06_case_loop.py 12
