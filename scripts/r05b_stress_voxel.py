"""round 5 (second sitting): seeded random stress of the sliding-window voxel-map kernel's round-5 extensions -- up to 64 grey
levels (table size / waves per workgroup templated) and the seventeen features of the WIDE instantiation -- against the
from-scratch window kernel (kernels_voxel.h, itself tied to the reference route by tests/test_gpu_configs.py): random shapes, level
counts 2..64, mask densities, radius 1 / 2, 3-D and force2D windows, random feature subsets, every voxel a centre.
usage: python scripts/r05b_stress_voxel.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyradiomics_amd import engine

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
ALL = ["JointEntropy", "JointEnergy", "JointAverage", "Autocorrelation", "ClusterProminence", "ClusterShade", "ClusterTendency", "Contrast",
       "DifferenceAverage", "DifferenceVariance", "Id", "Idm", "Idn", "Idmn", "InverseVariance", "SumAverage", "SumSquares"]
t0 = time.time()
n = {"slide": 0, "window": 0}
worst = {}
while time.time() - t0 < budget:
    shape = (int(rng.integers(1, 14)), int(rng.integers(3, 40)), int(rng.integers(3, 150)))
    Ng = int(rng.choice([2, 5, 16, 32, 33, 40, 41, 48, 49, 57, 64]))
    img = rng.integers(1, Ng + 1, size=shape).astype(np.int32)
    if rng.random() < 0.5:
        img = np.maximum(1, (img + int(rng.integers(1, 4))) // int(rng.integers(2, 5))).astype(np.int32)      # repeated pairs
    if rng.random() < 0.2:
        img[:] = int(rng.integers(1, Ng + 1))                                                                 # one level: zero variances
    msk = rng.random(shape) < float(rng.choice([1.0, 0.9, 0.5, 0.1]))
    if not msk.any():
        msk[0, 0, 0] = True
    vox = np.array(np.nonzero(np.ones(shape, bool))).astype(np.int32)
    kw = dict(kernelRadius=int(rng.choice([1, 2])))
    if shape[0] == 1 or rng.random() < 0.4:
        kw.update(force2D=True, force2Ddimension=0)
    k = int(rng.integers(1, len(ALL) + 1))
    feats = [ALL[i] for i in sorted(rng.choice(len(ALL), size=k, replace=False))]
    args = (torch.from_numpy(img).to(dev), torch.from_numpy(msk).to(dev), Ng, torch.from_numpy(vox).to(dev), feats)
    new = {f: v.cpu().numpy() for f, v in engine.voxel_glcm_features(*args, **kw).items()}
    variant = engine.last_variant()
    os.environ["PRAD_VOX_NO_SLIDE"] = "1"
    try:
        old = {f: v.cpu().numpy() for f, v in engine.voxel_glcm_features(*args, **kw).items()}
    finally:
        del os.environ["PRAD_VOX_NO_SLIDE"]
    for f in feats:
        a, b = new[f], old[f]
        assert np.array_equal(np.isnan(a), np.isnan(b)), (f, shape, Ng, kw)
        ok = ~np.isnan(a)
        if ok.any():
            err = float(np.max(np.abs(a[ok] - b[ok]) / (1e-10 + 1e-9 * np.abs(b[ok]))))
            worst[f] = max(worst.get(f, 0.0), err)
        np.testing.assert_allclose(a[ok], b[ok], rtol=1e-9, atol=1e-10, err_msg="%s %s Ng %d %s %s" % (f, shape, Ng, kw, feats))
    n[variant if variant in n else "window"] += 1
print("voxel stress ok: %s in %.0f s; worst |a-b| / (1e-10 + 1e-9 |b|) per feature: %s" % (n, time.time() - t0, {f: round(v, 4) for f, v in worst.items()}))
