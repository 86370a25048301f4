#!/bin/bash
# round 5: the same seeded stress on the product library and on the one-group build (is a mismatch the multi-group loop's?)
R=$(cd $(dirname $0)/.. && pwd)
cd $R
mkdir -p gpurun_out/r05g
timeout 300 python scripts/r05_stress.py ${1:-150} ${2:-5} > gpurun_out/r05g/stress_new.log 2>&1; tail -4 gpurun_out/r05g/stress_new.log | cut -c1-300
PRAD_LIB=$R/build_variants/lib_onegroup.so timeout 300 python scripts/r05_stress.py ${1:-150} ${2:-5} > gpurun_out/r05g/stress_onegroup.log 2>&1; tail -4 gpurun_out/r05g/stress_onegroup.log | cut -c1-300
PRAD_LIB=$R/build_variants/lib_oldlogic.so timeout 300 python scripts/r05_stress.py ${1:-150} ${2:-5} > gpurun_out/r05g/stress_oldlogic.log 2>&1; tail -4 gpurun_out/r05g/stress_oldlogic.log | cut -c1-300
