#!/bin/bash
run() { echo -n "$* : "; for d in uniform smooth; do env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dist $d --size ${SIZE:-512} 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | tr '\n' ' '; done; echo; }
for rs in 31 29 27 25 23 22 21 20 19 17 13; do run PRAD_FW_RS=$rs; done
