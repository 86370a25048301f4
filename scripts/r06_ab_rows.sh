cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" PRAD_BENCH_NOCHECK=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-modes --no-host-boundary 2>&1 | grep '^{"metric"' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'cold', d['cold_ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'))"; }
for rep in 1 2 3; do run V=base; run V=side PRAD_ROWS_STREAM=1; done


