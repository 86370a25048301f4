#!/bin/bash
run() { echo -n "$* : "; for d in uniform smooth; do env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-modes --no-host-boundary --dist $d --size ${SIZE:-512} 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | tr '\n' ' '; done; echo; }
for rs in 29 27 26 25 24 23 22 21 20 19 18 17 16; do run PRAD_FW_RS=$rs; done
