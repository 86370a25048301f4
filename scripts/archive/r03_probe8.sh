#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_fw.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
BA="--no-cpu-baseline --no-modes --no-host-boundary --steps 20 --warmup 3"
pick() { grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"rows_ms": [0-9.]*\|"pipeline_ms": [0-9.]*\|"pack_ms": [0-9.]*\|"sync_call_ms_per_step": [0-9.]*\|"sync_call_kernel_ms": {[^}]*}' | tr '\n' ' '; echo; }
echo "== pipeline uniform"; python bench.py $BA 2>&1 | tail -1 | pick
echo "== pipeline smooth"; python bench.py $BA --dist smooth 2>&1 | tail -1 | pick
echo "== lanes uniform"; python bench.py $BA --deferred-mode lanes 2>&1 | tail -1 | pick
echo "== lanes smooth"; python bench.py $BA --deferred-mode lanes --dist smooth 2>&1 | tail -1 | pick
echo "== 256"; python bench.py $BA --size 256 2>&1 | tail -1 | pick
