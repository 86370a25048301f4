#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
PRAD_FW_XCD=1 python -m pytest tests/test_gpu_fw.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
BA="--no-cpu-baseline --no-modes --no-host-boundary --steps 20 --warmup 3"
pick() { grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"rows_ms": [0-9.]*\|"sync_call_kernel_ms": {[^}]*}' | tr '\n' ' '; echo; }
for d in uniform smooth; do for x in 0 1; do
  echo "== xcd=$x $d"; PRAD_FW_XCD=$x python bench.py $BA --dist $d 2>&1 | tail -1 | pick
done; done
echo "== xcd=1 384"; PRAD_FW_XCD=1 python bench.py $BA --size 384 2>&1 | tail -1 | pick
echo "== xcd=0 384"; PRAD_FW_XCD=0 python bench.py $BA --size 384 2>&1 | tail -1 | pick
cd /tmp && export TMPDIR=/tmp
for x in 0 1; do
  PRAD_FW_XCD=$x rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/xcd_pmc$x -o f -- python $R/bench.py --steps 4 --warmup 2 $BA > /dev/null 2>&1
  echo "== FETCH_SIZE xcd=$x"; python $R/scripts/pmc_summary.py $R/gpurun_out/xcd_pmc$x sweep_fw_kernel | grep -E "###|FETCH"
done
