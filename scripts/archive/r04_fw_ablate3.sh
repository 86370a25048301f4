#!/bin/bash
# round 4: where the walk kernel's time goes beyond its bare step -- ablation builds (scripts/build_variant.sh), 512^3 uniform
R=$(cd $(dirname $0)/.. && pwd)
cd $R
run() { echo -n "$* : "; env "$@" PRAD_BENCH_NOCHECK=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-modes --no-host-boundary --dist ${DIST:-uniform} 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"kernel_ms_instrumented_pass": [0-9.]*' | tr '\n' ' '; echo; }
run V=base
for v in "$@"; do
  run V=$v PRAD_LIB=$R/build_variants/lib_$v.so
done
