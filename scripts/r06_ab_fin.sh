#!/bin/bash
# round 6: A/B of the finalize side stream (PRAD_FINALIZE_STREAM=1) on the headline loop, alternating runs on one box, + parity
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" PRAD_BENCH_NOCHECK=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-modes --no-host-boundary --levels ${LEVELS:-32} 2>&1 | grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'cold', d['cold_ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'))"; }
for rep in 1 2 3; do run V=base; run V=fin PRAD_FINALIZE_STREAM=1; done
LEVELS=64 run V=base; LEVELS=64 run V=fin PRAD_FINALIZE_STREAM=1
PRAD_FINALIZE_STREAM=1 python -m pytest tests/test_gpu_regress.py tests/test_gpu_fw.py tests/test_gpu_fw2_pack.py tests/test_gpu_stress.py tests/test_gpu_configs.py -q -m gpu -k "not c4 and not c5 and not c3" 2>&1 | tail -2
