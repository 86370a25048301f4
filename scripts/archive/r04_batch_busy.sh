#!/bin/bash
# round 4: is batch mode (bench.py modes.batch: 36 cases, three host threads) bound by the GPU or by its host threads?
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/b1.py <<PY
import sys, os; sys.path.insert(0, "$R")
import torch, time
import bench
dev = torch.device("cuda", 0)
nc, dt, nfeat = bench.mode_batch(dev, 0, 36, torch.cuda.synchronize)
print("batch: %.1f cases/s (%d cases, %.1f ms per case per GPU, one thread %.2f ms)" % (nc / dt, nc, dt / nc * 1e3, bench.mode_batch.one_thread_ms), flush=True)
PY
rm -rf /tmp/bb
PRAD_BATCH_THREADS=${1:-3} rocprofv3 --kernel-trace --stats -d /tmp/bb -o g -- python /tmp/b1.py 2>&1 | grep "batch:"
python $R/scripts/rocpd_busy.py /tmp/bb/g_results.db 0.5
