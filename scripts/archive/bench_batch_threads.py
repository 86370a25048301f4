#!/usr/bin/env python
"""cases/s of bench.py's batch mode for several numbers of host threads per GPU"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
def fence(): torch.cuda.synchronize()
for th in (1, 2, 3, 4, 6, 8):
    os.environ["PRAD_BATCH_THREADS"] = str(th)
    nc, dt, nf = bench.mode_batch(dev, 0, 16, fence)
    print("threads %d: %.1f cases/s (%.1f ms per case)" % (th, nc / dt, dt / nc * 1e3), flush=True)
