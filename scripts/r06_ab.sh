#!/bin/bash
# round 6: A/B of variant libraries (scripts/build_variant.sh) on the headline loop, alternating runs on one box.
#   usage: scripts/r06_ab.sh REPS variant [variant ...]      (DIST=smooth for the smooth volume, LEVELS=64 for the two-table walk)
R=$(cd $(dirname $0)/.. && pwd)
cd $R
O=gpurun_out/r06_ab
mkdir -p $O
REPS=$1; shift
run() { echo -n "$* : "; env "$@" PRAD_BENCH_NOCHECK=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-modes --no-host-boundary --dist ${DIST:-uniform} --levels ${LEVELS:-32} 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])"; }
for rep in $(seq 1 $REPS); do
  run V=base
  for v in "$@"; do
    run V=$v PRAD_LIB=$R/build_variants/lib_$v.so
  done
done 2>&1 | tee -a $O/ab_${DIST:-uniform}_${LEVELS:-32}.log
