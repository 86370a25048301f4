#!/usr/bin/env python
"""BASELINE config 4: voxel-based GLCM feature maps (exampleVoxel.yaml parameters: force2D, kernelRadius 2, masked
kernel, JointEntropy) of a synthetic volume, every voxel a kernel centre, fused on-device path.  With --gpus N
(torch.distributed launch) the centre list is split into contiguous slabs, one per rank, no collective."""
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
ap.add_argument("--levels", type=int, default=32)
ap.add_argument("--dist", default="smooth")
ap.add_argument("--radius", type=int, default=2)
ap.add_argument("--mode", choices=["2d", "3d"], default="2d")
ap.add_argument("--features", default="JointEntropy")
a = ap.parse_args()
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
img, msk = make_volume(a.size, a.levels, a.dist, 0, dev)
n = a.size
z0, z1 = (n * rank) // world, (n * (rank + 1)) // world      # this rank's slab of centres (whole volume resident)
zz, yy, xx = torch.meshgrid(torch.arange(z0, z1, device=dev, dtype=torch.int32),
                            torch.arange(n, device=dev, dtype=torch.int32),
                            torch.arange(n, device=dev, dtype=torch.int32), indexing="ij")
vox = torch.stack([zz.reshape(-1), yy.reshape(-1), xx.reshape(-1)])
feats = a.features.split(",")
kw = dict(kernelRadius=a.radius, force2D=(a.mode == "2d"), force2Ddimension=0)
engine.voxel_glcm_features(img, msk, a.levels, vox[:, :1000], feats, **kw)
torch.cuda.synchronize()
t = time.perf_counter()
res = engine.voxel_glcm_features(img, msk, a.levels, vox, feats, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t
nv = vox.shape[1]
print("rank %d/%d: %d^3 %s r=%d %s, %d kernels x %d feature(s): %.3f s = %.2f Mkernels/s (kernel %.1f ms), mean %s = %.6f"
      % (rank, world, n, a.mode, a.radius, a.dist, nv, len(feats), dt, nv / dt / 1e6, engine.last_kernel_ms("voxel"),
         feats[0], float(torch.nanmean(res[feats[0]]))), flush=True)
