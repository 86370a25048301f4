#!/bin/bash
R=$(cd $(dirname $0)/.. && pwd)
run() { echo -n "$* : "; env "$@" PRAD_BENCH_NOCHECK=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dist ${DIST:-uniform} --size ${SIZE:-512} 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; }
for v in base notail nodead notailnodead; do
  L=""; [ $v != base ] && L="PRAD_LIB=$R/build_variants/lib_$v.so"
  run V=$v $L
  run V=$v $L PRAD_FW_CL=32
done
