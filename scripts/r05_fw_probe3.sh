#!/bin/bash
R=$(cd $(dirname $0)/.. && pwd)
cd $R
O=gpurun_out/r05_fw3
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_fw.py tests/test_gpu_parity.py "tests/test_gpu_configs.py::test_headline_512_bit_exact" -x -q -m gpu -s 2>&1 | tail -25) | tee $O/tests.log
run() { echo -n "$* : "; env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-modes --no-host-boundary --dist ${DIST:-uniform} 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])"; }
for rep in 1 2; do
  run V=new
  run V=onegroup PRAD_LIB=$R/build_variants/lib_onegroup.so
  DIST=smooth run V=new
  DIST=smooth run V=onegroup PRAD_LIB=$R/build_variants/lib_onegroup.so
done 2>&1 | tee $O/ab.log
PRAD_LIB=$R/build_variants/lib_stamps.so python scripts/r05_fw_stamps.py uniform 2>&1 | tee $O/stamps_uniform.md
