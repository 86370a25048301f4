#!/bin/bash
# ablations of the fixed-window kernel (sweep family time of the bench, rows kernel included): base vs synthetic levels
# (no loads) vs all loads folded into 1 MB (L2-resident) vs no LDS atomic
R=$(cd $(dirname $0)/.. && pwd)
run() { echo -n "$* : "; env "$@" PRAD_BENCH_NOCHECK=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dist ${DIST:-uniform} --size ${SIZE:-512} 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; }
for v in base noload l2only nobump; do
  L=""; [ $v != base ] && L="PRAD_LIB=$R/build_variants/lib_$v.so"
  run V=$v $L
  run V=$v $L PRAD_FW_CL=512 PRAD_FW_BLOCKS=16
done
