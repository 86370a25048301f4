#!/bin/bash
run() { echo -n "$* : "; for d in uniform smooth; do env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-modes --no-host-boundary --dist $d --size ${SIZE:-512} 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | tr '\n' ' '; done; echo; }
for cl in 48 64 96 128 176 256; do run PRAD_FW_CL=$cl; done
for b in 20 21; do run PRAD_FW_BLOCKS=$b; done
