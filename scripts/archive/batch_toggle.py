#!/usr/bin/env python
"""mode_batch of bench.py (whole 256^3 cases) -- cases/s, for A/B runs with environment switches"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import mode_batch
dev = torch.device("cuda", 0)
def fence():
    torch.cuda.synchronize()
for _ in range(2):
    nc, dt, nf = mode_batch(dev, 0, 9, fence)
    print("%.2f cases/s (%.1f ms per case, %d features)" % (nc / dt, dt / nc * 1e3, nf), flush=True)
