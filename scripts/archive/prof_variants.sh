#!/bin/bash
# kernel-trace stats for selected ablation builds
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in "$@"; do
  PRAD_LIB=$R/build_variants/lib_$n.so PRAD_BENCH_NOCHECK=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pv_$n -o $n -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pv_$n.log 2>&1
  echo "== $n"; python $R/scripts/rocpd_stats.py $R/gpurun_out/pv_$n/${n}_results.db prad | head -8
done
