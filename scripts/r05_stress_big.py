"""round 5: the round-4 stress (scripts/r04_stress.py) on DEEP volumes -- the walk kernel's two bugs only showed beyond 128 slices.
GLSZM (dense tile model: many tiles, zones of millions of voxels and of one), GLDM / NGTDM (packed-byte kernel, many planes), the
deferred pipeline (the pack riding in the previous volume's walk) with slabs / plateaus / noise, partial masks.
usage: python scripts/r05_stress_big.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import binding
from pyradiomics_amd import cmatrices as cm, engine, _lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ck = binding.ref() if binding.have_ref() else binding.port()
rng = np.random.default_rng(seed)
n = {"glszm": 0, "neigh": 0, "pipeline": 0}
fails = []


def levels(shape, Ng, kind):
    if kind == "uniform":
        return rng.integers(1, Ng + 1, size=shape, dtype=np.int32)
    if kind == "slabs":
        ax = int(rng.integers(0, 3))
        ln = shape[ax]
        prof = np.empty(ln, np.int32)
        i = 0
        while i < ln:
            w = int(rng.integers(1, max(2, ln // 2)))
            prof[i:i + w] = rng.integers(1, Ng + 1)
            i += w
        sh = [1, 1, 1]
        sh[ax] = ln
        base = np.broadcast_to(prof.reshape(sh), shape).copy()
        noise = rng.random(shape) < rng.choice([0.0, 0.001, 0.02])
        base[noise] = rng.integers(1, Ng + 1, size=int(noise.sum()))
        return base
    f = rng.random(shape)
    for ax in range(3):
        f = f + np.roll(f, 1, ax) + np.roll(f, -1, ax) + (np.roll(f, 2, ax) if kind == "smooth2" else 0)
    if kind == "plateau":
        f = np.round(f * 2)
    f = (f - f.min()) / (np.ptp(f) + 1e-12)
    return np.minimum(Ng, 1 + np.floor(f * Ng)).astype(np.int32)


def mask_of(shape, kind):
    if kind == "full":
        return np.ones(shape, bool)
    if kind == "sparse":
        return rng.random(shape) < rng.choice([0.002, 0.02, 0.1])
    if kind == "ball":
        zz, yy, xx = np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing="ij")
        return (zz ** 2 + yy ** 2 + xx ** 2) < 0.8
    if kind == "bands":
        m = rng.random(shape) < 0.9
        for ax in range(3):
            idx = rng.random(shape[ax]) < 0.15
            sl = [slice(None)] * 3
            sl[ax] = idx
            m[tuple(sl)] = False
        return m
    return rng.random(shape) < rng.choice([0.5, 0.7, 0.95])


def bad(tag):
    fails.append(tag)
    print("MISMATCH " + tag, flush=True)


t0 = time.time()
while time.time() - t0 < budget * 0.4:
    Ng = int(rng.choice([1, 2, 8, 32, 64, 255]))
    shape = (int(rng.integers(1, 160)), int(rng.integers(1, 160)), int(rng.choice([1, 7, 8, 64, 65, 100, 128, 129, 200, 300, 512])))
    while shape[0] * shape[1] * shape[2] > 4_000_000:
        shape = (max(1, shape[0] // 2), shape[1], shape[2])
    img = levels(shape, Ng, rng.choice(["uniform", "smooth", "smooth2", "plateau", "slabs"]))
    mask = mask_of(shape, rng.choice(["full", "random", "sparse", "ball", "bands"]))
    if not mask.any():
        mask[0, 0, 0] = True
    f2 = bool(rng.random() < 0.2)
    dim = int(rng.integers(0, 3)) if f2 else 0
    Ns = int(mask.sum())
    want = ck.calculate_glszm(img, mask, Ng, Ns, f2, dim)
    got = cm.calculate_glszm(img, mask, Ng, Ns, f2, dim)
    if not (got.shape == want.shape and np.array_equal(got, want)):
        bad("GLSZM shape %s Ng %d force2D %s/%d path %s" % (shape, Ng, f2, dim, _lib.last_path()))
    n["glszm"] += 1

t1 = time.time()
while time.time() - t1 < budget * 0.3:
    Ng = int(rng.choice([2, 16, 32, 64, 128, 255]))
    shape = (int(rng.integers(1, 200)), int(rng.integers(1, 120)), int(rng.choice([4, 8, 36, 64, 100, 128, 232, 256, 300, 512])))
    while shape[0] * shape[1] * shape[2] > 6_000_000:
        shape = (max(1, shape[0] // 2), shape[1], shape[2])
    img = levels(shape, Ng, rng.choice(["uniform", "smooth", "plateau", "slabs"]))
    mask = mask_of(shape, rng.choice(["full", "random", "sparse", "ball", "bands"]))
    if not mask.any():
        mask[0, 0, 0] = True
    f2 = bool(rng.random() < 0.2)
    dim = int(rng.integers(0, 3)) if f2 else 0
    alpha = int(rng.choice([0, 0, 1, 3]))
    tag = "neigh shape %s Ng %d alpha %d force2D %s/%d" % (shape, Ng, alpha, f2, dim)
    if not np.array_equal(cm.calculate_gldm(img, mask, [1], Ng, alpha, f2, dim), ck.calculate_gldm(img, mask, [1], Ng, alpha, f2, dim)):
        bad("GLDM " + tag + " path " + _lib.last_path())
    got, en = cm.calculate_ngtdm(img, mask, [1], Ng, f2, dim), ck.calculate_ngtdm(img, mask, [1], Ng, f2, dim)
    if not (np.array_equal(got[..., 0], en[..., 0]) and np.allclose(got[..., 1:], en[..., 1:], rtol=1e-11, atol=0)):
        bad("NGTDM " + tag + " path " + _lib.last_path())
    n["neigh"] += 1

engine.set_deferred_mode(1)
t2 = time.time()
while time.time() - t2 < budget * 0.3:
    Ng = int(rng.choice([8, 32, 44]))
    shape = (int(rng.integers(40, 220)), int(rng.integers(8, 48)), int(rng.choice([128, 256, 512])))
    while shape[0] * shape[1] * shape[2] > 3_000_000:
        shape = (max(1, shape[0] * 2 // 3), shape[1], shape[2])
    vols = []
    for k in range(4):
        img = levels(shape, Ng, rng.choice(["uniform", "smooth", "plateau", "slabs", "slabs"]))
        mask = mask_of(shape, rng.choice(["full", "full", "random", "sparse", "bands"]))
        if not mask.any():
            mask[0, 0, 0] = True
        if rng.random() < 0.3:
            img = img.copy()
            img[~mask] = rng.choice(np.array([0, -3, 255, 256, 1 << 20, Ng + 1], dtype=np.int32), size=int((~mask).sum()))
        vols.append((img, mask))
    dev = [(torch.from_numpy(i).cuda(), torch.from_numpy(m.astype(np.uint8)).cuda()) for i, m in vols]
    Nr = max(shape)
    got = [engine.glcm_glrlm(i, m, Ng, Nr, deferred=True) for i, m in dev]
    engine.deferred_status()
    for k, ((img, mask), (g, r, _)) in enumerate(zip(vols, got)):
        eg, _a = ck.calculate_glcm(img, mask, [1], Ng, False, 0)
        er, _a = ck.calculate_glrlm(img, mask, Ng, Nr, False, 0)
        if not (np.array_equal(g.cpu().numpy(), eg.reshape(g.shape)) and np.array_equal(r.cpu().numpy(), er.reshape(r.shape))):
            bad("pipeline volume %d shape %s Ng %d" % (k, shape, Ng))
        n["pipeline"] += 1
engine.set_deferred_mode(-1)
print("big stress %s: %s cases in %.0f s (seed %d)%s" % ("ok" if not fails else "FAILED", n, time.time() - t0, seed, "" if not fails else " -- " + "; ".join(fails[:8])))
sys.exit(1 if fails else 0)
