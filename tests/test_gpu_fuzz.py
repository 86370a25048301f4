"""Seeded random sweep over the configuration space of the matrix calls: shapes (odd sizes, unit dimensions, rows
around the 16-byte pitch steps), grey-level counts on both sides of every kernel-path threshold (fused table <= 44,
byte levels <= 255, generic above), mask densities, smooth and random levels, force2D, distances.  Every case
compares the HIP path with the CPU oracle through the operator module, bit-exact (NGTDM float column: 1e-12 on these small volumes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NG_CHOICES = [1, 2, 3, 7, 16, 32, 33, 44, 45, 64, 128, 255, 256, 300]
DIM_CHOICES = [1, 2, 3, 4, 5, 7, 8, 13, 15, 16, 17, 24, 31, 32, 33, 40, 48, 63, 64, 65]


def _case(rng, deep=False):
    nd = int(rng.choice([2, 3, 3, 3, 3]))
    shape = tuple(int(rng.choice(DIM_CHOICES)) for _ in range(nd))
    while np.prod(shape) > 70000:
        shape = tuple(max(1, s // 2) for s in shape)
    Ng = int(rng.choice(NG_CHOICES))
    if deep:
        # a volume the fixed-window walk takes and cuts into >= 2 pieces per line (a march of >= 128 steps): dead piece starts,
        # tails, runs across piece boundaries -- the ground the two round-5 bugs of sweep_fw_kernel sat on for three rounds
        nd = 3
        shape = (int(rng.integers(130, 200)), int(rng.integers(9, 24)), int(rng.choice([96, 128, 200, 256])))
        if rng.random() < 0.4:
            shape = (shape[1], shape[0], shape[2])
        Ng = int(rng.choice([8, 32, 32, 44, 45, 64]))
    frac = float(rng.choice([1.0, 1.0, 0.9, 0.5, 0.08]))
    smooth = bool(rng.random() < 0.5)
    f = rng.random(shape)
    if smooth:
        for ax in range(nd):
            f = f + np.roll(f, 1, axis=ax) + np.roll(f, -1, axis=ax)
        if rng.random() < 0.3:       # plateaus: very long runs / big zones
            f = np.round(f * 1.5)
    span = np.ptp(f)
    used = max(1, min(Ng, int(rng.choice([Ng, Ng, max(1, Ng // 3)]))))     # sometimes only the low levels occur
    img = (1 + np.floor((f - f.min()) / (span + 1e-9) * used)).astype(np.int32) if span > 0 else np.ones(shape, np.int32)
    img = np.minimum(img, Ng)
    mask = np.ones(shape, bool) if frac >= 1.0 else rng.random(shape) < frac
    force2D = bool(rng.random() < 0.3)
    f2d = int(rng.integers(0, nd)) if force2D else 0
    dist = [[1], [1], [1], [2], [1, 2], [3]][int(rng.integers(0, 6))]
    alpha = int(rng.choice([0, 0, 1, 3]))
    return shape, Ng, img, mask, force2D, f2d, dist, alpha


@pytest.mark.parametrize("seed", range(8))
def test_random_configurations_match_oracle(seed, checker):
    from pyradiomics_amd import cmatrices as cm
    rng = np.random.default_rng(1000 + seed)
    from pyradiomics_amd import _lib
    for it in range(16):
        deep = it >= 14
        shape, Ng, img, mask, force2D, f2d, dist, alpha = _case(rng, deep)
        if deep:
            force2D, f2d, dist = False, 0, [1]
        tag = "seed %d it %d shape %s Ng %d force2D %s/%d dist %s alpha %d" % (seed, it, shape, Ng, force2D, f2d, dist, alpha)
        Nr = max(shape)
        try:
            want = checker.calculate_glcm(img, mask, dist, Ng, force2D, f2d)
        except (RuntimeError, IndexError) as e:          # e.g. no angle for this distance in this shape
            with pytest.raises(type(e)):
                cm.calculate_glcm(img, mask, dist, Ng, force2D, f2d)
            continue
        got = cm.calculate_glcm(img, mask, dist, Ng, force2D, f2d)
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0]), "GLCM " + tag
        want = checker.calculate_glrlm(img, mask, Ng, Nr, force2D, f2d)
        got = cm.calculate_glrlm(img, mask, Ng, Nr, force2D, f2d)
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0]), "GLRLM " + tag
        g, r, _ = cm.calculate_glcm_glrlm(img, mask, Ng, Nr, force2D, f2d)
        if deep:
            assert _lib.last_path() == "sweep" and _lib.last_variant() in ("fw", "fw2"), (tag, _lib.last_variant())
        assert np.array_equal(r, want[0]), "fused GLRLM " + tag
        assert np.array_equal(g, checker.calculate_glcm(img, mask, [1], Ng, force2D, f2d)[0]), "fused GLCM " + tag
        want = checker.calculate_gldm(img, mask, dist, Ng, alpha, force2D, f2d)
        assert np.array_equal(cm.calculate_gldm(img, mask, dist, Ng, alpha, force2D, f2d), want), "GLDM " + tag
        want = checker.calculate_ngtdm(img, mask, dist, Ng, force2D, f2d)
        got = cm.calculate_ngtdm(img, mask, dist, Ng, force2D, f2d)
        assert np.array_equal(got[..., [0, 2]], want[..., [0, 2]]), "NGTDM counts " + tag
        np.testing.assert_allclose(got[..., 1], want[..., 1], rtol=1e-12, atol=0, err_msg="NGTDM " + tag)
        Ns = int(mask.sum())
        if Ns:
            want = checker.calculate_glszm(img, mask, Ng, Ns, force2D, f2d)
            got = cm.calculate_glszm(img, mask, Ng, Ns, force2D, f2d)
            assert got.shape == want.shape and np.array_equal(got, want), "GLSZM " + tag
            Pc, sizes = cm.calculate_glszm_compact(img, mask, Ng, Ns, force2D, f2d)
            assert np.array_equal(Pc[0], want[0][:, sizes - 1]) and Pc.sum() == want.sum(), "GLSZM compact " + tag


@pytest.mark.parametrize("seed", range(3))
def test_random_voxel_batches_match_oracle(seed, checker):
    from pyradiomics_amd import cmatrices as cm
    rng = np.random.default_rng(2000 + seed)
    for it in range(6):
        shape = tuple(int(rng.choice([5, 8, 9, 12, 16])) for _ in range(3))
        Ng = int(rng.choice([3, 8, 32, 50]))
        img = rng.integers(1, Ng + 1, size=shape).astype(np.int32)
        mask = rng.random(shape) < float(rng.choice([1.0, 0.7]))
        radius = int(rng.choice([1, 2]))
        force2D = bool(rng.random() < 0.4)
        f2d = int(rng.integers(0, 3)) if force2D else 0
        pts = np.argwhere(mask)
        vox = pts[rng.choice(len(pts), size=min(40, len(pts)), replace=False)].T.astype(np.int32)
        kw = dict(kernelRadius=radius, voxels=vox)
        tag = "seed %d it %d shape %s Ng %d r %d force2D %s/%d" % (seed, it, shape, Ng, radius, force2D, f2d)
        want = checker.calculate_glcm(img, mask, [1], Ng, force2D, f2d, **kw)
        got = cm.calculate_glcm(img, mask, [1], Ng, force2D, f2d, **kw)
        assert np.array_equal(got[0], want[0]), "voxel GLCM " + tag
        want = checker.calculate_glrlm(img, mask, Ng, max(shape), force2D, f2d, **kw)
        got = cm.calculate_glrlm(img, mask, Ng, max(shape), force2D, f2d, **kw)
        assert np.array_equal(got[0], want[0]), "voxel GLRLM " + tag
        assert np.array_equal(cm.calculate_gldm(img, mask, [1], Ng, 0, force2D, f2d, **kw),
                              checker.calculate_gldm(img, mask, [1], Ng, 0, force2D, f2d, **kw)), "voxel GLDM " + tag
        assert np.array_equal(cm.calculate_ngtdm(img, mask, [1], Ng, force2D, f2d, **kw),
                              checker.calculate_ngtdm(img, mask, [1], Ng, force2D, f2d, **kw)), "voxel NGTDM " + tag
        Ns = int(mask.sum())
        assert np.array_equal(cm.calculate_glszm(img, mask, Ng, Ns, force2D, f2d, **kw),
                              checker.calculate_glszm(img, mask, Ng, Ns, force2D, f2d, **kw)), "voxel GLSZM " + tag


@pytest.mark.parametrize("seed", range(6))
def test_random_medium_volumes_match_oracle(seed, checker):
    """volumes big enough for the multi-line-per-lane sweeps (LPL 2 / 4), the 64x64 row tiles and many GLSZM tiles"""
    from pyradiomics_amd import cmatrices as cm
    rng = np.random.default_rng(3000 + seed)
    shape = tuple(int(rng.integers(40, 150)) for _ in range(3))
    if seed % 2:
        shape = shape[:2] + (int(rng.choice([64, 128, 192, 256, 272])),)
    Ng = int(rng.choice([8, 32, 44, 45, 100]))
    f = rng.random(shape)
    if seed % 3:
        for ax in range(3):
            f = f + np.roll(f, 1, axis=ax) + np.roll(f, 2, axis=ax) + np.roll(f, 3, axis=ax)
    img = np.minimum(1 + np.floor((f - f.min()) / (np.ptp(f) + 1e-9) * Ng), Ng).astype(np.int32)
    mask = np.ones(shape, bool) if seed % 2 else rng.random(shape) < 0.93
    tag = "seed %d shape %s Ng %d" % (seed, shape, Ng)
    Nr = max(shape)
    g, r, ang = cm.calculate_glcm_glrlm(img, mask, Ng, Nr, False, 0)
    want_g, want_ang = checker.calculate_glcm(img, mask, [1], Ng, False, 0)
    assert np.array_equal(ang, want_ang) and np.array_equal(g, want_g), "GLCM " + tag
    assert np.array_equal(r, checker.calculate_glrlm(img, mask, Ng, Nr, False, 0)[0]), "GLRLM " + tag
    assert np.array_equal(cm.calculate_gldm(img, mask, [1], Ng, 0, False, 0),
                          checker.calculate_gldm(img, mask, [1], Ng, 0, False, 0)), "GLDM " + tag
    got, want = cm.calculate_ngtdm(img, mask, [1], Ng, False, 0), checker.calculate_ngtdm(img, mask, [1], Ng, False, 0)
    assert np.array_equal(got[..., [0, 2]], want[..., [0, 2]]), "NGTDM counts " + tag
    # s_i is a rational number, sum_c S_c / c with integer S_c.  The HIP path evaluates exactly that (a few ulp);
    # the reference adds ~10^6 rounded terms in raster order and sits 1e-13 .. 1e-12 relative away from it.
    import torch
    from fractions import Fraction
    from oracle.segment_ops import OracleSegmentOps
    acc = OracleSegmentOps(checker).neigh_accumulate(1, torch.from_numpy(img), torch.from_numpy(mask.astype(np.uint8)),
                                                         Ng, 0, shape[0], 0, (1,), False, 0).numpy()
    exact = np.array([float(sum(Fraction(int(acc[g, c]), c) for c in range(1, acc.shape[1]))) for g in range(Ng)])
    np.testing.assert_allclose(got[0, :, 1], exact, rtol=1e-14, atol=0, err_msg="NGTDM vs exact " + tag)
    np.testing.assert_allclose(want[0, :, 1], exact, rtol=1e-10, atol=0, err_msg="reference NGTDM vs exact " + tag)
    Ns = int(mask.sum())
    want = checker.calculate_glszm(img, mask, Ng, Ns, False, 0)
    got = cm.calculate_glszm(img, mask, Ng, Ns, False, 0)
    assert got.shape == want.shape and np.array_equal(got, want), "GLSZM " + tag
