#!/bin/bash
# quick tuning sweep over lines-kernel launch parameters (env overrides read by prad_api.hip)
for lpl in 4 2 1; do for th in 1024 512; do
  for d in uniform smooth; do
    r=$(PRAD_LPL=$lpl PRAD_THREADS=$th timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --dist $d 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*')
    echo "LPL=$lpl threads=$th $d $r"
  done
done; done
