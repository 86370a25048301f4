#!/bin/bash
# times the bench (sweep kernels only) under several launch-geometry knobs of the fixed-window kernel
run() { echo -n "$* : "; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dist ${DIST:-uniform} --size ${SIZE:-512} 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*\|"pack_ms": [0-9.]*\|"pipeline_ms": [0-9.]*' | tr '\n' ' '; echo; }
run A=1
run PRAD_FW_CL=512
run PRAD_FW_CL=256
run PRAD_FW_CL=64
run PRAD_FW_CL=32
run PRAD_FW_PER_WAVE=3
run PRAD_FW_PER_WAVE=12
run PRAD_FW_BLOCKS=42
run PRAD_FW_BLOCKS=10
run PRAD_NO_FW=1
