#!/bin/bash
# times the bench with each ablation build in build_variants/ (see kernels_sweep.h PRAD_DBG_* / PRAD_LPL)
for f in build_variants/lib_*.so; do
  n=$(basename $f .so)
  for d in uniform smooth; do
    r=$(PRAD_LIB=$PWD/$f PRAD_BENCH_NOCHECK=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dist $d 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*')
    echo "$n $d $r"
  done
done
