#!/usr/bin/env python
"""BASELINE config 2: all five texture matrices of one synthetic volume on 1 MI355X, device-resident inputs.
Prints per-matrix wall ms (incl. the final sync each call ends with) and device ms (HIP events)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_volume
from pyradiomics_amd import engine

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--levels", type=int, default=32)
ap.add_argument("--dist", default="uniform")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
img, msk = make_volume(a.size, a.levels, a.dist, 0, dev)
n = img.numel()
jobs = {
    "glcm+glrlm": lambda: engine.glcm_glrlm(img, msk, a.levels, a.size),
    "gldm": lambda: engine.gldm(img, msk, a.levels),
    "ngtdm": lambda: engine.ngtdm(img, msk, a.levels),
    "glszm": lambda: engine.glszm(img, msk, a.levels, n),
    "glszm-compact": lambda: engine.glszm_compact(img, msk, a.levels, n),
    "firstorder": lambda: engine.firstorder_stats(img, msk, 0.0),
}
for name, fn in jobs.items():
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(a.reps):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    print("%-11s %dx%dx%d %s: %8.3f ms wall, %8.3f ms device (last call), %9.1f Mvox/s, path=%s"
          % (name, a.size, a.size, a.size, a.dist, best * 1e3, engine.last_device_ms(), n / best / 1e6,
             engine.last_path()), flush=True)
