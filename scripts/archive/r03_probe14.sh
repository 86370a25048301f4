#!/bin/bash
# grid size of the binning kernels (digitize: every workgroup flushes nb counters + its maximum with global atomics)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "4096 2048" "2048 1024" "1024 512" "512 256" "8192 4096"; do
  set -- $cfg
  rm -rf /tmp/pb
  PRAD_BIN_BLOCKS=$1 PRAD_MINMAX_BLOCKS=$2 rocprofv3 --kernel-trace --stats -d /tmp/pb -o s -- python $R/scripts/bench_binning.py > /tmp/pb.log 2>&1
  echo "== digitize blocks $1, minmax blocks $2"
  python $R/scripts/rocpd_stats.py /tmp/pb/s_results.db | grep -E "digitize|minmax"
done
