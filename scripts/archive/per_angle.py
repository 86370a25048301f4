#!/usr/bin/env python
"""sweep time of every angle on its own (whole GPU per angle) -- where the GLCM+GLRLM walk spends its time.
Usage: per_angle.py [N ...] [--dist uniform|smooth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_volume
from pyradiomics_amd import engine
dist = "smooth" if "--dist=smooth" in sys.argv else "uniform"
for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [256, 512]:
    img, msk = make_volume(n, 32, dist, 0, torch.device("cuda", 0))
    angles = engine.pair_angles(img.shape)
    for _ in range(3):
        engine.glcm_glrlm(img, msk, 32, n)
    allms = engine.last_kernel_ms("sweep")
    parts = []
    for a in angles:
        for _ in range(3):
            engine.glcm_glrlm(img, msk, 32, n, angles=a[None])
        parts.append(engine.last_kernel_ms("sweep"))
    print("N=%d %s: all 13 angles %.3f ms; alone: %s (sum %.3f)" % (
        n, dist, allms, " ".join("%s=%.3f" % ("".join("+0-"[int(v) if v >= 0 else 2] if False else str(int(v)) for v in a), p)
                                 for a, p in zip(angles, parts)), sum(parts)), flush=True)
