#!/usr/bin/env python
"""Maximum-size check of the headline path: a 1024 x 1024 x 1023 volume (2^30 - 2^20 voxels, just under the
reference's 2^31 - 1 element limit with headroom for the padded level volume) with a partial mask.  Verifies the
size-independent properties: per-angle GLCM pair totals against a direct count of masked neighbour pairs (torch),
GLRLM runs tile the ROI, symmetric-angle consistency, plus the non-power-of-two row length."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyradiomics_amd import engine

dev = torch.device("cuda", 0)
shape = (1024, 1024, 1020)
g = torch.Generator(device=dev); g.manual_seed(1)
img = torch.randint(1, 33, shape, generator=g, device=dev, dtype=torch.int32)
mask = (torch.rand(shape, generator=g, device=dev) < 0.97)
Ng, Nr = 32, 1024
t = time.perf_counter()
glcm, glrlm, ang = engine.glcm_glrlm(img, mask, Ng, Nr)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("%s voxels in %.2f ms (%.1f Gvox/s), path=%s" % (img.numel(), dt * 1e3, img.numel() / dt / 1e9, engine.last_path()))
assert engine.last_path() == "sweep"
nroi = int(mask.sum().item())
m = mask
for a in range(ang.shape[0]):
    dz, dy, dx = [int(v) for v in ang[a]]
    def sl(d, n):  # slices selecting p and p+d
        return (slice(0, n - d), slice(d, n)) if d >= 0 else (slice(-d, n), slice(0, n + d))
    (z0, z1), (y0, y1), (x0, x1) = sl(dz, shape[0]), sl(dy, shape[1]), sl(dx, shape[2])
    pairs = int((m[z0, y0, x0] & m[z1, y1, x1]).sum().item())
    assert int(glcm[:, :, a].sum().item()) == pairs, ("GLCM total", a)
    runs_vox = float((glrlm[:, :, a] * torch.arange(1, Nr + 1, device=dev, dtype=torch.float64)[None, :]).sum().item())
    assert runs_vox == float(nroi), ("GLRLM coverage", a, runs_vox, nroi)
    # pairs of equal level = sum (len-1) * runs
    same = int((m[z0, y0, x0] & m[z1, y1, x1] & (img[z0, y0, x0] == img[z1, y1, x1])).sum().item())
    diag = int(torch.diagonal(glcm[:, :, a]).sum().item())
    assert diag == same, ("GLCM diagonal", a, diag, same)
print("1024x1024x1020: GLCM totals, diagonals and GLRLM coverage of all %d angles verified; ROI %d voxels" % (ang.shape[0], nroi))

# the other matrices and the first-order statistics at the same size: identities that hold for exact results
Ngl = 32
counts = torch.bincount(img[mask], minlength=Ngl + 1)[1:].to(torch.float64)
t = time.perf_counter(); gldm = engine.gldm(img, mask, Ngl, 0); torch.cuda.synchronize(); t_gldm = time.perf_counter() - t
assert torch.equal(gldm.sum(1), counts)
t = time.perf_counter(); ngtdm = engine.ngtdm(img, mask, Ngl); torch.cuda.synchronize(); t_ngtdm = time.perf_counter() - t
assert torch.equal(ngtdm[:, 0], counts)
t = time.perf_counter(); P, sizes = engine.glszm_compact(img, mask, Ngl, nroi); torch.cuda.synchronize(); t_glszm = time.perf_counter() - t
assert torch.equal((P * torch.from_numpy(sizes).to(dev, torch.float64)[None, :]).sum(1), counts)
t = time.perf_counter(); st = engine.firstorder_stats(img, mask); torch.cuda.synchronize(); t_fo = time.perf_counter() - t
assert st["Np"] == nroi and st["Energy"] == float((counts * torch.arange(1, Ngl + 1, device=dev, dtype=torch.float64) ** 2).sum())
print("GLDM %.1f ms, NGTDM %.1f ms, GLSZM %.1f ms (%d zones, %d distinct sizes), first order %.1f ms: level totals, zone "
      "tiling and energy verified" % (t_gldm * 1e3, t_ngtdm * 1e3, t_glszm * 1e3, int(P.sum().item()), len(sizes), t_fo * 1e3))
