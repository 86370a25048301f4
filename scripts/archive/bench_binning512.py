"""binning kernels on a 512^3 float64 image (device ms per call through HIP events around the library call)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_volume
from pyradiomics_amd import engine
N = int(os.environ.get("BIN_N", "512"))
dev = torch.device("cuda", 0)
lv, msk = make_volume(N, 32, "smooth", 0, dev)
img = lv.to(torch.float64) * 25.0 + 3.0
for _ in range(2):
    engine.bin_image(img, msk, with_counts=True, binCount=32)
torch.cuda.synchronize()
ts = []
for _ in range(6):
    t0 = time.perf_counter(); engine.bin_image(img, msk, with_counts=True, binCount=32); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("bin_image %d^3 float64 (one-queue binCount): %.3f ms wall (min of 6)" % (N, min(ts)), flush=True)
