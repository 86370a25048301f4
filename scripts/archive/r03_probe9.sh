#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_fw.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
BA="--no-cpu-baseline --no-modes --no-host-boundary --steps 20 --warmup 3"
pick() { grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"rows_ms": [0-9.]*\|"pipeline_ms": [0-9.]*\|"sync_call_ms_per_step": [0-9.]*\|"sync_call_kernel_ms": {[^}]*}' | tr '\n' ' '; echo; }
for d in uniform smooth; do
echo "== plain tails $d"; python bench.py $BA --dist $d 2>&1 | tail -1 | pick
echo "== slow tails $d"; PRAD_LIB=$R/build_variants/lib_slowtail.so python bench.py $BA --dist $d 2>&1 | tail -1 | pick
for pw in 4 8 10; do echo "== plain tails per_wave=$pw $d"; PRAD_FW_PER_WAVE=$pw python bench.py $BA --dist $d 2>&1 | tail -1 | pick; done
done
