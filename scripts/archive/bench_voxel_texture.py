#!/usr/bin/env python
"""Fused voxel-based feature maps of GLRLM / GLSZM / GLDM / NGTDM: every voxel of an N^3 volume a kernel centre,
all features of the class.  Usage: bench_voxel_texture.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_volume
from pyradiomics_amd import cmatrices, engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
img, mask = make_volume(N, 32, "smooth", 0, dev)
vox = torch.nonzero(mask).T.to(torch.int32).contiguous()
for label, radius, f2d, frac in (("2-D 5x5 (exampleVoxel.yaml window)", 2, True, 1), ("3-D 3^3", 1, False, 1), ("3-D 5^3", 2, False, 8)):
    v = vox[:, :vox.shape[1] // frac].contiguous()
    for cls in ("gldm", "ngtdm", "glrlm", "glszm"):
        family, table = cmatrices._ZONE_LIKE[cls]
        ids = [i for i, f in enumerate(table) if f]
        engine.voxel_texture_features(family, img, mask, 32, v[:, :4096].contiguous(), ids, radius, f2d, 0)
        torch.cuda.synchronize(); t = time.perf_counter()
        out = engine.voxel_texture_features(family, img, mask, 32, v, ids, radius, f2d, 0)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("%-5s %d^3 %s, %d features: %d kernels in %.1f ms (%.1f M kernels/s), kernel %.1f ms"
              % (cls, N, label, len(ids), v.shape[1], dt * 1e3, v.shape[1] / dt / 1e6, engine.last_kernel_ms("voxel")), flush=True)
