#!/usr/bin/env python
"""Config 5 in miniature: whole-case extraction -- Original + 8 wavelet sub-bands, all five
texture classes -- on synthetic 256^3 volumes, timed per case for the device-resident route and for the host-array
route of the reference's call structure.  Usage: bench_cases.py [N] [cases] [smooth|uniform]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import make_volume
from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
from pyradiomics_amd.image import Image

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = sys.argv[3] if len(sys.argv) > 3 else "smooth"
params = {"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}}
mask = np.zeros((N, N, N), dtype=np.int16)
zz, yy, xx = np.ogrid[:N, :N, :N]
mask[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1      # a ball, ~38 % of the box
res = {}
for route in ("device", "host"):
    p = {k: dict(v) for k, v in params.items()}
    p["setting"]["deviceResident"] = route == "device"
    ex = RadiomicsFeatureExtractor(p)
    times = []
    for c in range((cases if route == "device" else 1) + 1):      # the host route takes ~10 s per case: one timed run
        vol = (make_volume(N, 32, kind, c, torch.device("cuda", 0))[0] * 25).cpu().numpy().astype(np.int16)
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = ex.execute(Image(vol), Image(mask))
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t)
        if c == 1:
            keep = out                      # the case both routes are compared on
    res[route] = (np.median(times[1:]), keep)
    print("%-6s route: %d^3 %s, %d features/case, median %.1f ms/case (%.2f cases/s, %.1f Mvox/s of ROI x 9 images)"
          % (route, N, kind, len(out), res[route][0] * 1e3, 1 / res[route][0],
             9 * int(mask.sum()) / res[route][0] / 1e6), flush=True)
def close(a, b):
    return a == b or (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-10 * abs(b)
same = all(close(float(res["device"][1][k]), float(res["host"][1][k])) for k in res["host"][1])
print("routes agree within 1e-10 relative (device-side formulas reorder float sums; matrices are bit-identical):", same)
