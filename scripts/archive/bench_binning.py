#!/usr/bin/env python
"""per-stage device times of the on-device discretisation of one 256^3 derived image (float64 wavelet band / float32 LoG)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_volume
from pyradiomics_amd import engine
dev = torch.device("cuda", 0)
lv, msk = make_volume(256, 32, "smooth", 0, dev)
img = (lv.to(torch.float32) * 25.0 + 3.0).to(torch.int16)
bands = engine.wavelet_images(img)
logi = engine.log_image(img, (1.0, 1.0, 1.0), 2.0)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
for name, d in (("wavelet-LLH f64", bands["wavelet-LLH"]), ("log f32", logi), ("original int16", img)):
    t_all = timeit(lambda: engine.bin_image(d, msk, with_counts=True, binCount=32))
    t_nocnt = timeit(lambda: engine.bin_image(d, msk, binCount=32))
    levels, Ng, _ = engine.bin_image(d, msk, binCount=32)
    t_cnt = timeit(lambda: engine.level_counts(levels, msk, Ng))
    print("%-16s bin_image+counts %.3f ms | bin_image %.3f ms | separate level_counts %.3f ms" % (name, t_all, t_nocnt, t_cnt), flush=True)
