#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_fw.py -x -q 2>&1 | tail -3
BA="--no-cpu-baseline --no-modes --no-host-boundary --steps 20 --warmup 3"
pick() { grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"rows_ms": [0-9.]*\|"pipeline_ms": [0-9.]*\|"pack_ms": [0-9.]*\|"sync_call_kernel_ms": {[^}]*}' | tr '\n' ' '; echo; }
echo "== current pipeline"; python bench.py $BA 2>&1 | tail -1 | pick
echo "== current no-inline"; PRAD_NO_INLINE_PACK=1 python bench.py $BA 2>&1 | tail -1 | pick
echo "== nopack variant (no-inline)"; PRAD_LIB=$R/build_variants/lib_nopack.so PRAD_NO_INLINE_PACK=1 python bench.py $BA 2>&1 | tail -1 | pick
for b in 1 2 3; do echo "== current every=$b"; PRAD_PACK_EVERY=$b python bench.py $BA 2>&1 | tail -1 | pick; done
echo "== smooth"; python bench.py $BA --dist smooth 2>&1 | tail -1 | pick
