#!/bin/bash
run() { echo -n "$* : "; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dist ${DIST:-uniform} --size ${SIZE:-512} 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; }
run A=1
run PRAD_FW_BUDGET_KB=72
run PRAD_FW_BUDGET_KB=72 PRAD_FW_BLOCKS=42
run PRAD_FW_BUDGET_KB=48 PRAD_FW_BLOCKS=63
run PRAD_FW_BUDGET_KB=100
