#!/bin/bash
R=$(cd $(dirname $0)/.. && pwd)
cd $R
O=gpurun_out/r05_fw
mkdir -p $O
PRAD_LIB=$R/build_variants/lib_stamps.so python scripts/r05_fw_stamps.py uniform 2>&1 | tee $O/stamps_uniform.md
PRAD_LIB=$R/build_variants/lib_stamps.so python scripts/r05_fw_stamps.py smooth 2>&1 | tee $O/stamps_smooth.md
PRAD_LIB=$R/build_variants/lib_l2only.so PRAD_BENCH_NOCHECK=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-modes --no-host-boundary 2>&1 | tail -5 | cut -c1-600 | tee $O/l2only.log
