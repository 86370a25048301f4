#!/bin/bash
# round 4: the walk at 256^3 under piece-length / workgroup-count overrides
R=$(cd $(dirname $0)/.. && pwd)
cd $R
run() { echo -n "$* : "; env "$@" python bench.py --steps 20 --warmup 3 --size 256 --no-cpu-baseline --no-modes --no-host-boundary --dist ${DIST:-uniform} 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"sync_call_kernel_ms": {[^}]*}' | tr '\n' ' '; echo; }
for v in "$@"; do run V=1 $v; done
