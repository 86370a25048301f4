#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
BA="--no-cpu-baseline --no-modes --no-host-boundary --steps 20 --warmup 3"
pick() { grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"rows_ms": [0-9.]*\|"pipeline_ms": [0-9.]*\|"pack_ms": [0-9.]*\|"sync_call_ms_per_step": [0-9.]*\|"sync_call_kernel_ms": {[^}]*}' | tr '\n' ' '; echo; }
for ng in 32 44 45 64 96 128 160; do for n in 256 512; do for d in uniform smooth; do
  echo "== Ng=$ng size=$n $d"; python bench.py $BA --levels $ng --size $n --dist $d 2>&1 | tail -1 | pick
done; done; done
echo "== Ng=64 512 uniform NO_FW2 (r02 path)"; PRAD_NO_FW2=1 python bench.py $BA --levels 64 2>&1 | tail -1 | pick
echo "== Ng=128 512 uniform NO_FW2 (r02 path)"; PRAD_NO_FW2=1 python bench.py $BA --levels 128 2>&1 | tail -3 | cut -c1-300
