"""round 5 (second sitting): seeded random stress of what this sitting changed, against the reference C (oracle/_ref) or our C port:
  * the deferred pipeline with the two-table walk's side-job pack (PackWave16): batches of 2-5 deferred volumes of one shape or of
    mixed shapes / level counts (fused-table and two-table volumes alternate: the side job only rides in a launch of its own kind),
    rows that are / are not whole 16-voxel pieces, full / random / sparse / banded masks (the three conversion paths of a piece),
    junk levels outside the mask;
  * the x angle of a two-table volume on the 16-bit levels (sweep_fw2_rows_kernel) and the prefetching x-angle kernel at <= 44
    levels: long runs / plateaus / constant slabs along x, 45..160 levels, rows of 65..512 voxels, > 4096 rows and fewer.
usage: python scripts/r05b_stress.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import binding
from pyradiomics_amd import engine, _lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ck = binding.ref() if binding.have_ref() else binding.port()
rng = np.random.default_rng(seed)
n = {"batches": 0, "volumes": 0, "fw2": 0, "fw": 0, "other": 0}
fails = []


def levels(shape, Ng, kind):
    if kind == "uniform":
        return rng.integers(1, Ng + 1, size=shape, dtype=np.int32)
    if kind == "xslabs":                      # constant stretches of random length along x: runs of tens to hundreds
        ln = shape[2]
        base = np.empty(shape, np.int32)
        for z in range(shape[0]):
            prof = np.empty(ln, np.int32)
            i = 0
            while i < ln:
                w = int(rng.integers(1, max(2, ln // 3)))
                prof[i:i + w] = rng.integers(1, Ng + 1)
                i += w
            base[z] = prof[None, :]
        noise = rng.random(shape) < 0.03
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
    if kind == "bands":
        m = rng.random(shape) < 0.9
        for ax in range(3):
            idx = rng.random(shape[ax]) < 0.15
            sl = [slice(None)] * 3
            sl[ax] = idx
            m[tuple(sl)] = False
        return m
    if kind == "box":                         # a box ROI: pieces fully inside, fully outside and across its faces
        m = np.zeros(shape, bool)
        lo = [int(rng.integers(0, s // 2)) for s in shape]
        hi = [int(rng.integers(l + 1, s + 1)) for l, s in zip(lo, shape)]
        m[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = True
        return m
    return rng.random(shape) < rng.choice([0.5, 0.7, 0.95])


def rand_shape():
    nx = int(rng.choice([128, 200, 256, 296, 300, 320, 511, 512, 512, 72, 80]))
    shape = (int(rng.integers(6, 120)), int(rng.integers(6, 70)), nx)
    if rng.random() < 0.3:
        shape = (shape[1], shape[0], nx)
    while np.prod(shape) > 2_500_000:
        shape = (shape[0] // 2 + 5, shape[1] // 2 + 5, nx)
    return shape


t0 = time.time()
engine.set_deferred_mode(1)
while time.time() - t0 < budget:
    nvol = int(rng.integers(2, 6))
    mixed = rng.random() < 0.35
    shape0, Ng0 = rand_shape(), int(rng.choice([45, 64, 64, 100, 129, 160, 32, 16]))
    vols = []
    for i in range(nvol):
        shape = rand_shape() if (mixed and rng.random() < 0.5) else shape0
        Ng = int(rng.choice([32, 64, 100, 24, 160])) if (mixed and rng.random() < 0.5) else Ng0
        img = levels(shape, Ng, rng.choice(["uniform", "smooth", "smooth2", "plateau", "xslabs", "xslabs"]))
        mask = mask_of(shape, rng.choice(["full", "full", "random", "sparse", "bands", "box", "box"]))
        if not mask.any():
            mask[0, 0, 0] = True
        if rng.random() < 0.3:                # junk outside the mask: ignored, like the reference (cmatrices.c:61-64)
            out = ~mask
            img = img.copy()
            img[out] = rng.choice(np.array([0, -5, 255, 256, 1 << 20, -(1 << 30), 65535, 65536, 32768, Ng + 1], dtype=np.int32), size=int(out.sum()))
        vols.append((img, mask, Ng))
    dev = [(torch.from_numpy(i).cuda(), torch.from_numpy(m.astype(np.uint8)).cuda()) for i, m, _ in vols]
    got, variants = [], []
    for (di, dm), (_, _, Ng) in zip(dev, vols):
        got.append(engine.glcm_glrlm(di, dm, Ng, 512, deferred=True))
        variants.append(engine.last_variant())
    engine.deferred_status()
    for (img, mask, Ng), (g, r, _), var in zip(vols, got, variants):
        wg, _ = ck.calculate_glcm(img, mask, [1], Ng, False, 0)
        wr, _ = ck.calculate_glrlm(img, mask, Ng, 512, False, 0)
        g, r = g.cpu().numpy(), r.cpu().numpy()
        okg, okr = np.array_equal(g, wg[0]), np.array_equal(r, wr[0])
        n["volumes"] += 1
        n[var if var in ("fw", "fw2") else "other"] += 1
        if not (okg and okr):
            tag = "batch %d shape %s Ng %d variant %s (batch: %s)" % (n["batches"], img.shape, Ng, var, [(v[0].shape, v[2]) for v in vols])
            fails.append(tag)
            bad_a = sorted(set(np.argwhere(g != wg[0])[:, -1].tolist()) | set(np.argwhere(r != wr[0])[:, -1].tolist()))
            print("MISMATCH %s glcm %s glrlm %s angles %s |dG| %g |dR| %g" % (tag, okg, okr, bad_a, np.abs(g - wg[0]).sum(), np.abs(r - wr[0]).sum()), flush=True)
            if len(fails) <= 3:
                out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r05b_stress_fail_%d.npz" % len(fails))
                np.savez_compressed(out, **{"img%d" % j: v[0] for j, v in enumerate(vols)}, **{"mask%d" % j: v[1] for j, v in enumerate(vols)},
                                    Ng=np.array([v[2] for v in vols]))
    n["batches"] += 1
engine.set_deferred_mode(-1)
print("stress %s: %s in %.0f s (seed %d)%s" % ("ok" if not fails else "FAILED", n, time.time() - t0, seed, "" if not fails else " -- " + "; ".join(fails[:5])))
sys.exit(1 if fails else 0)
