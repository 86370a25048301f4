#!/bin/bash
# round 4: the walk kernel under tuning overrides (environment), 512^3; usage: r04_fw_env.sh "VAR=a VAR2=b" "VAR=c" ...
R=$(cd $(dirname $0)/.. && pwd)
cd $R
run() { echo -n "$* : "; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-modes --no-host-boundary --dist ${DIST:-uniform} 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"kernel_ms_instrumented_pass": [0-9.]*' | tr '\n' ' '; echo; }
for v in "$@"; do
  run V=1 $v
done
