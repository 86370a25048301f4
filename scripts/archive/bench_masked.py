#!/usr/bin/env python
"""GLCM+GLRLM sweep time with a partial ROI (ball / random mask): the lines that run through unmasked voxels"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_volume
from pyradiomics_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
for dist in ("uniform", "smooth"):
    img, full = make_volume(n, 32, dist, 0, dev)
    ax = torch.linspace(-1, 1, n, device=dev)
    ball = ((ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2) < 0.8).to(torch.uint8)
    rnd = (torch.rand((n, n, n), device=dev) < 0.7).to(torch.uint8)
    for name, m in (("full", full), ("ball (42 %)", ball), ("random 70 %", rnd)):
        for _ in range(3):
            engine.glcm_glrlm(img, m, 32, n)
        print("%d^3 %-8s mask %-12s: pack %.3f  sweep %.3f  total %.3f ms" % (n, dist, name, engine.last_kernel_ms("pack"),
              engine.last_kernel_ms("sweep"), engine.last_device_ms()), flush=True)
