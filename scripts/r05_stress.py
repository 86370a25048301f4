"""round 5: seeded random stress of what this round changed, against the reference C (oracle/_ref) or our C port:
  fw     the fixed-window walk with several plain groups per margin check (PRAD_FW_MAXG): LONG marches (up to 300 steps along z
         and y), long runs / plateaus / constant slabs, sparse and banded masks, rows of 128..512 voxels, 8..44 levels
  pairs  the pairs tier: GLCM for distances [1], [2], [1,2], [1,2,3] (2-D), GLCM + GLRLM at 161..1200 levels, GLDM / NGTDM at
         256..600 levels, 2-D and 3-D, force2D, isolated voxels (the 2-D-angle rule)
usage: python scripts/r05_stress.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding
from pyradiomics_amd import cmatrices as cm, _lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ck = binding.ref() if binding.have_ref() else binding.port()
rng = np.random.default_rng(seed)
n = {"fw": 0, "pairs_glcm": 0, "pairs_runs": 0, "pairs_neigh": 0}
fails = []


def levels(shape, Ng, kind):
    if kind == "uniform":
        return rng.integers(1, Ng + 1, size=shape, dtype=np.int32)
    if kind == "slabs":                       # constant stretches of random length along a random axis: runs of tens to hundreds
        ax = int(rng.integers(0, len(shape)))
        ln = shape[ax]
        prof = np.empty(ln, np.int32)
        i = 0
        while i < ln:
            w = int(rng.integers(1, max(2, ln // 2)))
            prof[i:i + w] = rng.integers(1, Ng + 1)
            i += w
        sh = [1] * len(shape)
        sh[ax] = ln
        base = np.broadcast_to(prof.reshape(sh), shape).copy()
        noise = rng.random(shape) < 0.02
        base[noise] = rng.integers(1, Ng + 1, size=int(noise.sum()))
        return base
    f = rng.random(shape)
    for ax in range(len(shape)):
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
    if kind == "bands":                        # whole rows / planes outside the ROI: the row flags, calm_zero
        m = rng.random(shape) < 0.9
        for ax in range(len(shape)):
            idx = rng.random(shape[ax]) < 0.15
            sl = [slice(None)] * len(shape)
            sl[ax] = idx
            m[tuple(sl)] = False
        return m
    return rng.random(shape) < rng.choice([0.5, 0.7, 0.95])


t0 = time.time()
while time.time() - t0 < budget * 0.6:
    Ng = int(rng.choice([8, 16, 32, 33, 44, 45, 64, 100, 160]))        # (45+: the two-table walk)
    nx = int(rng.choice([128, 200, 256, 300, 511, 512, 512, 600, 1000, 1024]))        # (513..1024: the K = 16 window)
    shape = (int(rng.integers(20, 300)), int(rng.integers(9, 60)), nx)
    if rng.random() < 0.3:
        shape = (shape[1], shape[0], nx)
    if np.prod(shape) > 5_000_000:
        shape = (shape[0] // 2 + 9, shape[1] // 2 + 9, nx)
    img = levels(shape, Ng, rng.choice(["uniform", "smooth", "smooth2", "plateau", "slabs", "slabs"]))
    mask = mask_of(shape, rng.choice(["full", "full", "random", "sparse", "bands"]))
    if not mask.any():
        mask[0, 0, 0] = True
    Nr = max(shape)
    f2 = bool(rng.random() < 0.2)
    dim = int(rng.integers(0, 3)) if f2 else 0
    g, r, ang = cm.calculate_glcm_glrlm(img, mask, Ng, Nr, f2, dim)
    tag = "fw shape %s Ng %d force2D %s/%d variant %s" % (shape, Ng, f2, dim, _lib.last_variant() if hasattr(_lib, "last_variant") else "?")
    assert _lib.last_path() in ("sweep", "pairs"), tag      # (45+ levels on rows beyond 512 voxels: the pairs tier)
    wg, wang = ck.calculate_glcm(img, mask, [1], Ng, f2, dim)
    wr, _ = ck.calculate_glrlm(img, mask, Ng, Nr, f2, dim)
    okg, okr = np.array_equal(ang, wang) and np.array_equal(g, wg), np.array_equal(r, wr)
    if not (okg and okr):
        fails.append(tag)
        bad_a = sorted(set(np.argwhere(g != wg)[:, -1].tolist()) | set(np.argwhere(r != wr)[:, -1].tolist()))
        print("MISMATCH %s glcm %s glrlm %s angles %s |dG| %g |dR| %g" % (tag, okg, okr, bad_a, np.abs(g - wg).sum(), np.abs(r - wr).sum()), flush=True)
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r05_stress_fail_%d.npz" % len(fails))
        if len(fails) <= 3:
            np.savez_compressed(out, img=img.astype(np.int16), mask=mask, Ng=Ng)
    n["fw"] += 1

t1 = time.time()
while time.time() - t1 < budget * 0.4:
    nd = 3 if rng.random() < 0.8 else 2
    shape = tuple(int(rng.integers(4, 40)) for _ in range(nd))
    which = rng.choice(["glcm", "runs", "neigh"])
    f2 = bool(nd == 3 and rng.random() < 0.25)
    dim = int(rng.integers(0, 3)) if f2 else 0
    mask = mask_of(shape, rng.choice(["full", "random", "sparse", "bands"]))
    if not mask.any():
        mask[(0,) * nd] = True
    if which == "glcm":
        Ng = int(rng.choice([5, 32, 64, 200, 300, 700]))
        dist = [[1], [2], [1, 2], [1, 2, 3]][int(rng.integers(0, 4 if nd == 2 else 3))]
        if dist == [1] and Ng <= 160:
            dist = [1, 2]
        img = levels(shape, Ng, rng.choice(["uniform", "smooth", "plateau", "slabs"]))
        g, ang = cm.calculate_glcm(img, mask, dist, Ng, f2, dim)
        assert _lib.last_path() == "pairs", ("glcm", shape, Ng, dist, _lib.last_path())
        wg, wang = ck.calculate_glcm(img, mask, dist, Ng, f2, dim)
        assert np.array_equal(ang, wang) and np.array_equal(g, wg), ("glcm", shape, Ng, dist, f2, dim)
        n["pairs_glcm"] += 1
    elif which == "runs":
        Ng = int(rng.choice([161, 200, 255, 256, 400, 1200]))
        img = levels(shape, Ng, rng.choice(["uniform", "smooth", "plateau", "slabs", "slabs"]))
        if rng.random() < 0.5:
            img = (img + 19) // 20                                    # few distinct levels under a large Ng: long runs
        Nr = max(shape)
        g, r, ang = cm.calculate_glcm_glrlm(img, mask, Ng, Nr, f2, dim)
        assert _lib.last_path() == "pairs", ("runs", shape, Ng, _lib.last_path())
        wg, wang = ck.calculate_glcm(img, mask, [1], Ng, f2, dim)
        wr, _ = ck.calculate_glrlm(img, mask, Ng, Nr, f2, dim)
        assert np.array_equal(ang, wang) and np.array_equal(g, wg) and np.array_equal(r, wr), ("runs", shape, Ng, f2, dim)
        n["pairs_runs"] += 1
    else:
        Ng = int(rng.choice([256, 300, 600]))
        dist = [[1], [1, 2]][int(rng.integers(0, 2))]
        alpha = int(rng.choice([0, 1, 5]))
        img = levels(shape, Ng, rng.choice(["uniform", "smooth", "plateau"]))
        a = cm.calculate_gldm(img, mask, dist, Ng, alpha, f2, dim)
        assert _lib.last_path() == "pairs", ("gldm", shape, Ng, dist, _lib.last_path())
        assert np.array_equal(a, ck.calculate_gldm(img, mask, dist, Ng, alpha, f2, dim)), ("gldm", shape, Ng, dist, alpha, f2, dim)
        b = cm.calculate_ngtdm(img, mask, dist, Ng, f2, dim)
        assert _lib.last_path() == "pairs"
        w = ck.calculate_ngtdm(img, mask, dist, Ng, f2, dim)
        assert np.array_equal(b[..., 0], w[..., 0]) and np.array_equal(b[..., 2], w[..., 2]), ("ngtdm counts", shape, Ng, dist)
        assert np.allclose(b[..., 1], w[..., 1], rtol=1e-12, atol=1e-12), ("ngtdm sums", shape, Ng, dist)
        n["pairs_neigh"] += 1
print("stress %s: %s cases in %.0f s (seed %d)%s" % ("ok" if not fails else "FAILED", n, time.time() - t0, seed, "" if not fails else " -- " + "; ".join(fails)))
sys.exit(1 if fails else 0)
