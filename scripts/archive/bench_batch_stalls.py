#!/usr/bin/env python
"""per-case latencies inside bench.py's batch mode with T threads: where does the run-to-run spread come from?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pyradiomics_amd import batch
from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
from pyradiomics_amd.image import Image
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
T = int(os.environ.get("BATCH_T", "3")); n = int(os.environ.get("BATCH_N", "36"))
N = 256
zz, yy, xx = np.ogrid[:N, :N, :N]
roi = np.zeros((N, N, N), dtype=np.int16)
roi[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1
ex = RadiomicsFeatureExtractor({"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}})
vols = [(bench.make_volume(N, 32, "smooth", c, dev)[0] * 25).cpu().numpy().astype(np.int16) for c in range(n + 1)]
lat = []
def one(c):
    t0 = time.perf_counter()
    r = ex.execute(Image(vols[c]), Image(roi))
    lat.append((time.perf_counter() - t0) * 1e3)
    return r
one(0)
batch.warm_threads(lambda: one(0), T)
for rep in range(3):
    lat.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    batch._run_threaded(list(range(n)), list(range(1, n + 1)), one, T)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    l = sorted(lat)
    st = torch.cuda.memory_stats()
    print("threads %d rep %d: %.1f cases/s; per-case ms min %.1f median %.1f p90 %.1f max %.1f; torch mallocs so far %d, reserved %.1f GB"
          % (T, rep, n / dt, l[0], l[len(l) // 2], l[int(len(l) * 0.9)], l[-1], st.get("num_device_alloc", -1),
             torch.cuda.memory_reserved() / 2 ** 30), flush=True)
