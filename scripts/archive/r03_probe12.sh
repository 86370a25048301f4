#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
BA="--no-cpu-baseline --no-modes --no-host-boundary --steps 20 --warmup 3"
pick() { grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"sync_call_kernel_ms": {[^}]*}' | tr '\n' ' '; echo; }
echo "== xcd=0 default"; PRAD_FW_XCD=0 python bench.py $BA 2>&1 | tail -1 | pick
echo "== xcd=0 CL=64"; PRAD_FW_XCD=0 PRAD_FW_CL=64 python bench.py $BA 2>&1 | tail -1 | pick
echo "== xcd=1 (CL=64)"; PRAD_FW_XCD=1 python bench.py $BA 2>&1 | tail -1 | pick
echo "== xcd=1 CL=128"; PRAD_FW_XCD=1 PRAD_FW_CL=128 python bench.py $BA 2>&1 | tail -1 | pick
echo "== xcd=1 CL=88"; PRAD_FW_XCD=1 PRAD_FW_CL=88 python bench.py $BA 2>&1 | tail -1 | pick
