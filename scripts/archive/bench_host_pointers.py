#!/usr/bin/env python
"""PCIe-inclusive rate of the drop-in boundary: `cmatrices.calculate_glcm_glrlm` on HOST numpy arrays (pageable int32
volume + uint8 mask in, float64 matrices out), the way radiomics.cMatrices is called.  Usage: bench_host_pointers.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyradiomics_amd import cmatrices
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
img = np.random.default_rng(0).integers(1, 33, (N, N, N), dtype=np.int32)
mask = np.ones((N, N, N), dtype=np.uint8)
cmatrices.calculate_glcm_glrlm(img[:8], mask[:8], 32, N, False, 0)
best = 1e9
for _ in range(3):
    t = time.perf_counter()
    g, r, _ = cmatrices.calculate_glcm_glrlm(img, mask, 32, N, False, 0)
    best = min(best, time.perf_counter() - t)
print("host-pointer GLCM+GLRLM %d^3: %.1f ms per call = %.0f Mvoxels/s (%.1f GB/s of pageable input)" % (
    N, best * 1e3, N ** 3 / best / 1e6, 5 * N ** 3 / best / 1e9), flush=True)
