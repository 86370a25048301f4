#!/usr/bin/env python
"""BASELINE config 3: wavelet (8 sub-bands) + LoG (sigma 1..5 mm) of a synthetic volume, each derived image
re-discretised (binCount 32) and pushed through the GLCM+GLRLM build, everything resident on one MI355X."""
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
a = ap.parse_args()
dev = torch.device("cuda", 0)
lv, msk = make_volume(a.size, 32, "smooth", 0, dev)
img = (lv.to(torch.float32) * 25.0 + 3.0).to(torch.int16)       # an intensity image with the same structure
n = img.numel()


def run():
    t = {}
    t0 = time.perf_counter()
    derived = engine.wavelet_images(img)
    torch.cuda.synchronize(); t["wavelet x8"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for s in (1.0, 2.0, 3.0, 4.0, 5.0):
        derived["log-sigma-%g" % s] = engine.log_image(img, (1.0, 1.0, 1.0), s)
    torch.cuda.synchronize(); t["LoG x5"] = time.perf_counter() - t0
    tb = tm = 0.0
    for name, d in derived.items():
        t0 = time.perf_counter()
        levels, Ng, _ = engine.bin_image(d, msk, binCount=32)
        torch.cuda.synchronize(); tb += time.perf_counter() - t0
        t0 = time.perf_counter()
        engine.glcm_glrlm(levels, msk, Ng, a.size)
        torch.cuda.synchronize(); tm += time.perf_counter() - t0
        assert engine.last_path() == "sweep"
    t["binning x13"] = tb
    t["GLCM+GLRLM x13"] = tm
    return t


run()
t = run()
tot = sum(t.values())
print("%d^3, 13 derived images: " % a.size + ", ".join("%s %.1f ms" % (k, v * 1e3) for k, v in t.items())
      + " | total %.1f ms = %.1f Mvox/s of derived volume" % (tot * 1e3, 13 * n / tot / 1e6))
