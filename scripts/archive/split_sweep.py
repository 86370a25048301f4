#!/usr/bin/env python
"""pack / sweep / finalize split of the GLCM+GLRLM call for a few volume sizes.  Usage: split_sweep.py N [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_volume
from pyradiomics_amd import engine
for n in [int(a) for a in sys.argv[1:]] or [231, 256]:
    img, msk = make_volume(n, 32, "uniform", 0, torch.device("cuda", 0))
    for _ in range(3):
        engine.glcm_glrlm(img, msk, 32, n)
    print("N=%d: total %.3f ms = pack %.3f + sweep %.3f + finalize %.3f" % (n, engine.last_device_ms(), engine.last_kernel_ms("pack"),
          engine.last_kernel_ms("sweep"), engine.last_kernel_ms("finalize")))
