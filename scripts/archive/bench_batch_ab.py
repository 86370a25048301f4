#!/usr/bin/env python
"""cases/s of bench.py's batch mode, N cases, T threads (env BATCH_N, BATCH_T), repeated"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
def fence(): torch.cuda.synchronize()
n = int(os.environ.get("BATCH_N", "48"))
for th in [int(t) for t in os.environ.get("BATCH_T", "6").split(",")]:
    os.environ["PRAD_BATCH_THREADS"] = str(th)
    r = []
    for _ in range(3):
        nc, dt, nf = bench.mode_batch(dev, 0, n, fence)
        r.append(nc / dt)
    print("enqueue=%s threads %d, %d cases: %s cases/s" % (os.environ.get("PRAD_ENQUEUE_SEGMENT", "1"), th, n, " ".join("%.1f" % x for x in r)), flush=True)
