"""Seeded randomised stress in the driver-run suite: bounded slices of the generators that found the two
walk-kernel bugs of rounds 2-4 (scripts/r05_stress.py, r05b_stress.py, r04_stress.py, r05b_stress_voxel.py), every case against
the reference's own C (oracle/_ref) or our restatement of it.  What tests/test_gpu_fuzz.py cannot reach with volumes of 70 000
voxels: marches of 128 .. 300 steps (a walk is cut into pieces: dead starts, tails), constant slabs whose thickness sits around the
table's last length slot, runs that end at the x edge of window-filling rows (Nx = 256 / 512 / 1024), banded masks, the deferred
pipeline with fused-table and two-table volumes back to back, GLSZM / GLDM / NGTDM on ragged shapes, and the sliding-window voxel
kernel against the REFERENCE route (per-kernel matrices of the reference C + the numpy formulas of glcm.py).
Proof that it finds what it is for: the test builds -DPRAD_DBG_R5BUG1 / -DPRAD_DBG_R5BUG2 (the walk as it was before the round-5
fixes, scripts/build_variant.sh) fail test_fw_long_marches on the seeds below (profiles/r06_probes.md section 5)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _levels(rng, shape, Ng, kind):
    if kind == "uniform":
        return rng.integers(1, Ng + 1, size=shape, dtype=np.int32)
    if kind in ("slabs", "slabs1"):
        # constant stretches along one axis: runs of tens to hundreds of voxels, thickness often right around the fused table's last
        # length slot (20 .. 30 at 32 levels); slabs1: every third slab is level 1 (the level whose dead lines the margin mis-read)
        ax = int(rng.integers(0, len(shape)))
        ln = shape[ax]
        prof = np.empty(ln, np.int32)
        i = 0
        while i < ln:
            w = int(rng.integers(16, 34)) if rng.random() < 0.6 else int(rng.integers(1, max(2, ln // 2)))
            prof[i:i + w] = 1 if (kind == "slabs1" and rng.random() < 0.35) else rng.integers(1, Ng + 1)
            i += w
        sh = [1] * len(shape)
        sh[ax] = ln
        base = np.broadcast_to(prof.reshape(sh), shape).copy()
        noise = rng.random(shape) < 0.02
        base[noise] = rng.integers(1, Ng + 1, size=int(noise.sum()))
        return base
    if kind == "xslabs":
        ln = shape[-1]
        base = np.empty(shape, np.int32)
        for z in range(shape[0]):
            prof = np.empty(ln, np.int32)
            i = 0
            while i < ln:
                w = int(rng.integers(1, max(2, ln // 3)))
                prof[i:i + w] = rng.integers(1, Ng + 1)
                i += w
            base[z] = prof
        noise = rng.random(shape) < 0.03
        base[noise] = rng.integers(1, Ng + 1, size=int(noise.sum()))
        return base
    f = rng.random(shape)
    for ax in range(len(shape)):
        f = f + np.roll(f, 1, ax) + np.roll(f, -1, ax) + (np.roll(f, 2, ax) if kind == "smooth2" else 0)
    if kind == "plateau":
        f = np.round(f * 2)
    f = (f - f.min()) / (np.ptp(f) + 1e-12)
    return np.minimum(Ng, 1 + np.floor(f * Ng)).astype(np.int32)


def _mask(rng, shape, kind):
    if kind == "full":
        return np.ones(shape, bool)
    if kind == "sparse":
        m = rng.random(shape) < rng.choice([0.002, 0.02, 0.1])
    elif kind == "bands":                      # whole rows / planes outside the ROI: the row flags, calm_zero
        m = rng.random(shape) < 0.9
        for ax in range(len(shape)):
            idx = rng.random(shape[ax]) < 0.15
            sl = [slice(None)] * len(shape)
            sl[ax] = idx
            m[tuple(sl)] = False
    elif kind == "box":
        m = np.zeros(shape, bool)
        lo = [int(rng.integers(0, max(1, s // 2))) for s in shape]
        hi = [int(rng.integers(l + 1, s + 1)) for l, s in zip(lo, shape)]
        m[tuple(slice(l, h) for l, h in zip(lo, hi))] = True
    elif kind == "ball":
        g = np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing="ij")
        m = sum(x ** 2 for x in g) < 0.8
    else:
        m = rng.random(shape) < rng.choice([0.5, 0.7, 0.95])
    if not m.any():
        m[(0,) * len(shape)] = True
    return m


JUNK = np.array([0, -5, 255, 256, 1 << 20, -(1 << 30), 65535, 65536, 32768], dtype=np.int32)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_fw_long_marches(seed, checker):
    """the fixed-window walks (fused table <= 44 levels, two tables above) on marches long enough to be cut into pieces"""
    from pyradiomics_amd import cmatrices as cm, _lib
    rng = np.random.default_rng(600 + seed)
    deep = 0
    for it in range(16):
        Ng = int(rng.choice([8, 16, 32, 32, 32, 33, 44, 45, 64, 100]))
        nx = int(rng.choice([128, 256, 256, 300, 512, 512, 1024]))
        nz, ny = int(rng.integers(128, 300)), int(rng.integers(9, 40))
        if rng.random() < 0.3:
            nz, ny = ny, nz
        while nz * ny * nx > 2_400_000:
            if ny > 12:
                ny = ny // 2 + 5
            else:
                nz = nz * 3 // 4
        shape = (nz, ny, nx)
        img = _levels(rng, shape, Ng, rng.choice(["uniform", "smooth", "plateau", "slabs", "slabs", "slabs1", "slabs1"]))
        mask = _mask(rng, shape, rng.choice(["full", "full", "full", "random", "sparse", "bands"]))
        Nr = max(shape)
        f2 = bool(rng.random() < 0.15)
        dim = int(rng.integers(0, 3)) if f2 else 0
        g, r, ang = cm.calculate_glcm_glrlm(img, mask, Ng, Nr, f2, dim)
        variant = _lib.last_variant()
        tag = "seed %d it %d shape %s Ng %d force2D %s/%d variant %s" % (seed, it, shape, Ng, f2, dim, variant)
        assert _lib.last_path() in ("sweep", "pairs"), tag
        if variant in ("fw", "fw2") and max(nz, ny) >= 128:
            deep += 1
        wg, wang = checker.calculate_glcm(img, mask, [1], Ng, f2, dim)
        wr, _ = checker.calculate_glrlm(img, mask, Ng, Nr, f2, dim)
        assert np.array_equal(ang, wang), tag
        bad = sorted(set(np.argwhere(g != wg)[:, -1].tolist()) | set(np.argwhere(r != wr)[:, -1].tolist()))
        assert not bad, "%s: angles %s differ, |dGLCM| %g |dGLRLM| %g" % (tag, bad, np.abs(g - wg).sum(), np.abs(r - wr).sum())
    assert deep >= 10, "only %d of 16 volumes took a fixed-window walk with a march of >= 128 steps" % deep


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_deferred_pipeline_mixed_batches(seed, checker):
    """batches of deferred volumes, fused-table and two-table kinds back to back: the pack that rides in the previous volume's walk"""
    import torch
    from pyradiomics_amd import engine
    rng = np.random.default_rng(700 + seed)

    def rand_shape():
        nx = int(rng.choice([128, 200, 256, 296, 320, 511, 512, 72]))
        shape = (int(rng.integers(6, 140)), int(rng.integers(6, 60)), nx)
        if rng.random() < 0.3:
            shape = (shape[1], shape[0], nx)
        while np.prod(shape) > 1_300_000:
            shape = (shape[0] // 2 + 5, shape[1] // 2 + 5, nx)
        return shape

    engine.set_deferred_mode(1)
    seen = set()
    try:
        for b in range(6):
            nvol = int(rng.integers(2, 5))
            mixed = rng.random() < 0.4
            shape0, Ng0 = rand_shape(), int(rng.choice([45, 64, 64, 100, 160, 32, 32, 16]))
            vols = []
            for _ in range(nvol):
                shape = rand_shape() if (mixed and rng.random() < 0.5) else shape0
                Ng = int(rng.choice([32, 64, 100, 24, 160])) if (mixed and rng.random() < 0.5) else Ng0
                img = _levels(rng, shape, Ng, rng.choice(["uniform", "smooth", "plateau", "xslabs", "slabs"]))
                mask = _mask(rng, shape, rng.choice(["full", "full", "random", "sparse", "bands", "box"]))
                if rng.random() < 0.3:            # junk outside the mask: ignored, like the reference (cmatrices.c:61-64)
                    img = img.copy()
                    img[~mask] = rng.choice(np.append(JUNK, Ng + 1).astype(np.int32), size=int((~mask).sum()))
                vols.append((img, mask, Ng))
            dev = [(torch.from_numpy(i).cuda(), torch.from_numpy(m.astype(np.uint8)).cuda()) for i, m, _ in vols]
            got, variants = [], []
            for (di, dm), (_, _, Ng) in zip(dev, vols):
                got.append(engine.glcm_glrlm(di, dm, Ng, 512, deferred=True))
                variants.append(engine.last_variant())
            engine.deferred_status()
            for (img, mask, Ng), (g, r, _), var in zip(vols, got, variants):
                seen.add(var)
                wg, _ = checker.calculate_glcm(img, mask, [1], Ng, False, 0)
                wr, _ = checker.calculate_glrlm(img, mask, Ng, 512, False, 0)
                tag = "seed %d batch %d shape %s Ng %d variant %s (batch: %s)" % (seed, b, img.shape, Ng, var, [(v[0].shape, v[2]) for v in vols])
                assert np.array_equal(g.cpu().numpy(), wg[0]), "GLCM " + tag
                assert np.array_equal(r.cpu().numpy(), wr[0]), "GLRLM " + tag
    finally:
        engine.set_deferred_mode(-1)
    assert {"fw", "fw2"} <= seen, seen


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_zones_and_neighbourhoods_ragged(seed, checker):
    """GLSZM (dense tile-root labelling), GLDM / NGTDM (packed-byte neighbourhood kernel) on ragged shapes and masks"""
    from pyradiomics_amd import cmatrices as cm
    rng = np.random.default_rng(800 + seed)
    n = 0
    for it in range(36):
        f2 = bool(rng.random() < 0.3)
        dim = int(rng.integers(0, 3)) if f2 else 0
        if it % 2 == 0:
            Ng = int(rng.choice([1, 2, 3, 8, 32, 64, 200, 255]))
            shape = (int(rng.integers(1, 70)), int(rng.integers(1, 70)), int(rng.choice([1, 3, 4, 7, 8, 9, 16, 31, 64, 65, 100, 128, 129, 200, 300])))
            while np.prod(shape) > 400000:
                shape = (max(1, shape[0] // 2), shape[1], shape[2])
            img = _levels(rng, shape, Ng, rng.choice(["uniform", "smooth", "smooth2", "plateau"]))
            mask = _mask(rng, shape, rng.choice(["full", "random", "sparse", "ball"]))
            Ns = int(mask.sum())
            try:
                want = checker.calculate_glszm(img, mask, Ng, Ns, f2, dim)
            except (RuntimeError, IndexError):
                continue
            got = cm.calculate_glszm(img, mask, Ng, Ns, f2, dim)
            assert got.shape == want.shape and np.array_equal(got, want), "GLSZM seed %d it %d shape %s Ng %d force2D %s/%d" % (seed, it, shape, Ng, f2, dim)
        else:
            Ng = int(rng.choice([2, 16, 32, 64, 127, 128, 200, 255]))
            shape = (int(rng.integers(1, 30)), int(rng.integers(1, 40)), int(rng.choice([4, 8, 12, 36, 64, 100, 128, 232, 256])))
            img = _levels(rng, shape, Ng, rng.choice(["uniform", "smooth", "plateau"]))
            mask = _mask(rng, shape, rng.choice(["full", "random", "sparse", "ball"]))
            dist = [[1], [1], [1, 2]][int(rng.integers(0, 3))]
            alpha = int(rng.choice([0, 0, 1, 3]))
            tag = "seed %d it %d shape %s Ng %d dist %s alpha %d force2D %s/%d" % (seed, it, shape, Ng, dist, alpha, f2, dim)
            try:
                ed = checker.calculate_gldm(img, mask, dist, Ng, alpha, f2, dim)
            except (RuntimeError, IndexError):
                continue
            en = checker.calculate_ngtdm(img, mask, dist, Ng, f2, dim)
            assert np.array_equal(cm.calculate_gldm(img, mask, dist, Ng, alpha, f2, dim), ed), "GLDM " + tag
            got = cm.calculate_ngtdm(img, mask, dist, Ng, f2, dim)
            assert np.array_equal(got[..., 0], en[..., 0]) and np.array_equal(got[..., 2], en[..., 2]), "NGTDM counts " + tag
            assert np.allclose(got[..., 1], en[..., 1], rtol=1e-12, atol=0), "NGTDM sums " + tag
        n += 1
    assert n >= 24


SLIDE_FEATS = ["JointEntropy", "JointEnergy", "JointAverage", "Autocorrelation", "ClusterProminence", "ClusterShade", "ClusterTendency", "Contrast",
               "DifferenceAverage", "DifferenceVariance", "Id", "Idm", "Idn", "Idmn", "InverseVariance", "SumAverage", "SumSquares"]


def _reference_route(checker, img, msk, Ng, vox, force2D, radius):
    """voxel-based GLCM features the way the reference computes them: per-kernel matrices from the C checker (_cmatrices.c:203-222,
    set_bb :1120-1147), then glcm.py:149-205 (symmetrise, delete the angles that are empty for EVERY kernel, NaN where a kernel has no
    pair on an angle, normalise) and the formulas (:260-887): nanmean over the angles, JointAverage the plain mean"""
    import warnings
    P, _ = checker.calculate_glcm(img, msk, [1], Ng, force2D, 0, kernelRadius=radius, voxels=np.ascontiguousarray(vox))
    P = P + P.transpose(0, 2, 1, 3)
    tot = P.sum((1, 2))
    keep = tot.sum(0) > 0
    P, tot = P[..., keep], tot[:, keep]
    tot[tot == 0] = np.nan
    lev = np.arange(1, Ng + 1, dtype=float)
    I, J = lev[None, :, None, None], lev[None, None, :, None]
    K = np.abs(I - J)
    out = {}
    with np.errstate(invalid="ignore", divide="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        p = P / tot[:, None, None, :]

        def nm(w):
            return np.nanmean((p * w).sum((1, 2)), 1)
        ux = (I * p).sum((1, 2), keepdims=True)
        c = I + J - 2 * ux
        da = (p * K).sum((1, 2), keepdims=True)
        out["JointEntropy"] = np.nanmean(-(p * np.log2(p + np.spacing(1))).sum((1, 2)), 1)
        out["JointEnergy"] = np.nanmean((p ** 2).sum((1, 2)), 1)
        out["JointAverage"] = ux[:, 0, 0, :].mean(1)
        out["Autocorrelation"] = nm(I * J)
        out["ClusterProminence"] = nm(c ** 4)
        out["ClusterShade"] = nm(c ** 3)
        out["ClusterTendency"] = nm(c ** 2)
        out["Contrast"] = nm((I - J) ** 2)
        out["DifferenceAverage"] = np.nanmean(da[:, 0, 0, :], 1)
        out["DifferenceVariance"] = nm((K - da) ** 2)
        out["Id"] = nm(1.0 / (1.0 + K))
        out["Idm"] = nm(1.0 / (1.0 + K ** 2))
        out["Idn"] = nm(1.0 / (1.0 + K / Ng))
        out["Idmn"] = nm(1.0 / (1.0 + K ** 2 / Ng ** 2))
        out["InverseVariance"] = nm(np.where(K > 0, 1.0 / np.where(K > 0, K, 1.0) ** 2, 0.0))
        out["SumAverage"] = nm(I + J)
        out["SumSquares"] = nm((I - ux) ** 2)
    return out


@pytest.mark.parametrize("seed", [0, 1])
def test_voxel_slide_kernel_vs_reference_route(seed, checker):
    """the sliding-window voxel kernel (every voxel a centre, <= 64 levels, 17 features) against the reference route"""
    import torch
    from pyradiomics_amd import engine
    rng = np.random.default_rng(900 + seed)
    dev = torch.device("cuda", 0)
    slide = 0
    for it in range(8):
        shape = (int(rng.integers(1, 7)), int(rng.integers(6, 20)), int(rng.integers(6, 40)))
        Ng = int(rng.choice([2, 5, 16, 32, 33, 41, 49, 64]))
        img = rng.integers(1, Ng + 1, size=shape).astype(np.int32)
        if rng.random() < 0.5:
            img = np.maximum(1, (img + int(rng.integers(1, 4))) // int(rng.integers(2, 5))).astype(np.int32)      # repeated pairs
        msk = rng.random(shape) < float(rng.choice([1.0, 0.9, 0.5]))
        if not msk.any():
            msk[0, 0, 0] = True
        vox = np.array(np.nonzero(np.ones(shape, bool))).astype(np.int32)
        radius = int(rng.choice([1, 2]))
        f2 = bool(shape[0] == 1 or rng.random() < 0.4)
        kw = dict(kernelRadius=radius, force2D=f2, force2Ddimension=0) if f2 else dict(kernelRadius=radius)
        k = int(rng.integers(3, len(SLIDE_FEATS) + 1))
        feats = [SLIDE_FEATS[i] for i in sorted(rng.choice(len(SLIDE_FEATS), size=k, replace=False))]
        got = engine.voxel_glcm_features(torch.from_numpy(img).to(dev), torch.from_numpy(msk).to(dev), Ng, torch.from_numpy(vox).to(dev), feats, **kw)
        slide += engine.last_variant() == "slide"
        want = _reference_route(checker, img, msk, Ng, vox, f2, radius)
        for f in feats:
            a, b = got[f].cpu().numpy(), want[f]
            tag = "%s seed %d it %d shape %s Ng %d %s" % (f, seed, it, shape, Ng, kw)
            assert np.array_equal(np.isnan(a), np.isnan(b)), tag
            ok = ~np.isnan(b)
            np.testing.assert_allclose(a[ok], b[ok], rtol=1e-9, atol=1e-10, err_msg=tag)
    assert slide >= 4, "only %d of 8 requests took the sliding-window kernel" % slide
