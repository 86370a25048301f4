#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_fw.py -x -q 2>&1 | tail -3
BA="--no-cpu-baseline --no-modes --no-host-boundary --steps 20 --warmup 3"
pick() { grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"rows_ms": [0-9.]*\|"pipeline_ms": [0-9.]*\|"pack_ms": [0-9.]*\|"finalize_ms": [0-9.]*\|"sync_call_ms_per_step": [0-9.]*\|"sync_call_kernel_ms": {[^}]*}' | tr '\n' ' '; echo; }
for d in uniform smooth; do
  echo "== pipeline dist=$d"; python bench.py $BA --dist $d 2>&1 | tail -1 | pick
  echo "== pipeline no-inline-pack dist=$d"; PRAD_NO_INLINE_PACK=1 python bench.py $BA --dist $d 2>&1 | tail -1 | pick
  for b in 2 4 12 40; do echo "== pipeline burst=$b dist=$d"; PRAD_PACK_BURST=$b python bench.py $BA --dist $d 2>&1 | tail -1 | pick; done
done
