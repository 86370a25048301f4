#!/bin/bash
# round 4: the 64-level path (two-table walk + x-angle kernel + 16-bit pack), per-family device ms of the synchronous call
R=$(cd $(dirname $0)/.. && pwd)
cd $R
run() { echo -n "$* : "; env "$@" PRAD_BENCH_NOCHECK=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-modes --no-host-boundary --levels 64 --dist ${DIST:-uniform} 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"sync_call_kernel_ms": {[^}]*}' | tr '\n' ' '; echo; }
run V=base
for v in "$@"; do
  if [ -f $R/build_variants/lib_$v.so ]; then run V=$v PRAD_LIB=$R/build_variants/lib_$v.so; else run V=1 $v; fi
done
