#!/usr/bin/env python
"""First-order statistics throughput: segment mode on N^3 (all 15 statistics behind the 19 features) and the
voxel-based maps (2-D 5x5 and 3-D 5^3 kernels, all 19 features / Entropy only).  Usage: bench_firstorder.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyradiomics_amd import engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
for dtype in (torch.int16, torch.float64):
    img = (torch.randn((N, N, N), generator=g, device=dev) * 300 + 800).to(dtype)
    mask = torch.ones((N, N, N), dtype=torch.uint8, device=dev)
    for _ in range(2):
        engine.firstorder_stats(img, mask, 0.0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        st = engine.firstorder_stats(img, mask, 0.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print("segment %d^3 %s: %.2f ms wall (%.1f Gvox/s), kernels %.2f ms" % (N, str(dtype).split(".")[-1], dt * 1e3, N ** 3 / dt / 1e9,
                                                                     engine.last_kernel_ms("firstorder")), flush=True)
img = (torch.randn((N, N, N), generator=g, device=dev) * 300 + 800).to(torch.int16)
mask = torch.ones((N, N, N), dtype=torch.uint8, device=dev)
levels, Ng, _ = engine.bin_image(img, mask, binWidth=25)
vox = torch.nonzero(mask).T.to(torch.int32).contiguous()
for label, radius, f2d, ids in (("2-D 5x5, 19 features", 2, True, list(range(19))), ("2-D 5x5, Entropy", 2, True, [2]),
                                ("3-D 3^3, 19 features", 1, False, list(range(19))), ("3-D 5^3, Entropy", 2, False, [2])):
    n = vox.shape[1] if radius == 1 or f2d else vox.shape[1] // 8
    v = vox[:, :n].contiguous()
    engine.voxel_firstorder(img, mask, levels, v[:, :1000].contiguous(), ids, radius, f2d, 0)
    torch.cuda.synchronize(); t = time.perf_counter()
    out = engine.voxel_firstorder(img, mask, levels, v, ids, radius, f2d, 0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("voxel map %d^3 %s: %d kernels in %.1f ms (%.1f M kernels/s)" % (N, label, n, dt * 1e3, n / dt / 1e6), flush=True)
