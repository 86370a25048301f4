#!/bin/bash
# round 5: is the walk kernel held up by its loads?  A/B of the variant libraries (scripts/build_variant.sh) + the phase table
R=$(cd $(dirname $0)/.. && pwd)
cd $R
O=gpurun_out/r05_fw
mkdir -p $O
run() { echo -n "$* : "; env "$@" PRAD_BENCH_NOCHECK=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-modes --no-host-boundary --dist ${DIST:-uniform} 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])"; }
for rep in 1 2; do
  run V=base
  for v in "$@"; do
    run V=$v PRAD_LIB=$R/build_variants/lib_$v.so
  done
done 2>&1 | tee $O/ab.log
PRAD_LIB=$R/build_variants/lib_stamps.so python scripts/r05_fw_stamps.py uniform 2>&1 | tee $O/stamps_uniform.md
PRAD_LIB=$R/build_variants/lib_stamps.so python scripts/r05_fw_stamps.py smooth 2>&1 | tee $O/stamps_smooth.md
if [ -f $R/build_variants/lib_stampslate.so ]; then
  PRAD_LIB=$R/build_variants/lib_stampslate.so python scripts/r05_fw_stamps.py uniform 2>&1 | tee $O/stampslate_uniform.md
fi
