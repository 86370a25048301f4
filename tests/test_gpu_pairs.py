"""The tier between the sweeps and the exact generic kernels (kernels_pairs.h, round 5): GLCM for any `distances`
(cmatrices.c:4-92 with the angle list of :807-892) and GLCM / GLRLM above 160 grey levels (:299-541), bit for bit against
the reference C (oracle/_ref when it travelled with the repo, else the pinned C restatement)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(shape, Ng, seed, smooth=False, hole=True):
    rng = np.random.default_rng(seed)
    if smooth:
        from scipy import ndimage
        f = ndimage.gaussian_filter(rng.standard_normal(shape), 1.5)
        img = (1 + (f - f.min()) / (f.max() - f.min()) * (Ng - 1e-9)).astype(np.int32)
    else:
        img = rng.integers(1, Ng + 1, size=shape).astype(np.int32)
    msk = rng.random(shape) < 0.85
    if hole and len(shape) == 3:
        msk[:, shape[1] // 3: shape[1] // 3 + 3, :] = False
    return img, msk


@pytest.mark.parametrize("shape,Ng,dist,smooth", [((40, 37, 52), 32, [1, 2], False), ((24, 30, 33), 32, [2], True),
                                                   ((18, 20, 70), 7, [1, 2], True), ((12, 64, 64), 64, [1, 2], False),
                                                   ((30, 33), 16, [1, 2, 3], True), ((9, 10, 11), 300, [1, 2], False)])
def test_glcm_any_distances_takes_the_pairs_tier(shape, Ng, dist, smooth, checker):
    from pyradiomics_amd import cmatrices as cm, _lib
    img, msk = _case(shape, Ng, 3, smooth)
    got, ang = cm.calculate_glcm(img, msk, dist, Ng, False, 0)
    assert _lib.last_path() == "pairs", _lib.last_path()
    want, wang = checker.calculate_glcm(img, msk, dist, Ng, False, 0)
    assert np.array_equal(ang, wang) and np.array_equal(got, want)


@pytest.mark.parametrize("shape,Ng,smooth", [((40, 37, 52), 300, False), ((33, 48, 40), 200, True), ((20, 70, 66), 255, False),
                                              ((64, 64, 64), 1000, True), ((31, 45), 400, True)])
def test_glcm_glrlm_above_160_levels_take_the_pairs_tier(shape, Ng, smooth, checker):
    from pyradiomics_amd import cmatrices as cm, _lib
    img, msk = _case(shape, Ng, 5, smooth)
    if smooth:                       # long runs and a constant slab: the run walk, N_g - sum len * runs
        img[..., : shape[-1] // 3] = (img[..., : shape[-1] // 3] + 39) // 40
    Nr = int(max(shape))
    g, r, ang = cm.calculate_glcm_glrlm(img, msk, Ng, Nr, False, 0)
    assert _lib.last_path() == "pairs", _lib.last_path()
    wg, wang = checker.calculate_glcm(img, msk, [1], Ng, False, 0)
    wr, _ = checker.calculate_glrlm(img, msk, Ng, Nr, False, 0)
    assert np.array_equal(ang, wang)
    assert np.array_equal(g, wg), "GLCM"
    assert np.array_equal(r, wr), "GLRLM"
    r2, _ = cm.calculate_glrlm(img, msk, Ng, Nr, False, 0)
    assert np.array_equal(r2, wr), "GLRLM alone"


def test_glrlm_two_d_angle_rule_and_force2d(checker):
    """cmatrices.c:524-534: an angle none of whose lines holds two ROI voxels loses its length-1 column -- isolated ROI voxels
    (never two on a line of the z angles), two ROI voxels on a line that are NOT neighbours, force2D"""
    from pyradiomics_amd import cmatrices as cm, _lib
    Ng, shape = 300, (6, 20, 24)
    rng = np.random.default_rng(2)
    img = rng.integers(1, Ng + 1, size=shape).astype(np.int32)
    one_slice = np.zeros(shape, bool)
    one_slice[2] = rng.random(shape[1:]) < 0.7                       # a 2-D segmentation inside a 3-D array
    gap = one_slice.copy()
    gap[5, 3, 4] = gap[2, 3, 4] = True                               # same (0,0,1)... line along z, two voxels, not adjacent
    gap[3, 3, 4] = gap[4, 3, 4] = False
    for msk in (one_slice, gap):
        for f2d in (False, True):
            r, ang = cm.calculate_glrlm(img, msk, Ng, max(shape), f2d, 0)
            assert _lib.last_path() == "pairs"
            wr, wang = checker.calculate_glrlm(img, msk, Ng, max(shape), f2d, 0)
            assert np.array_equal(ang, wang) and np.array_equal(r, wr)
            g, _ = cm.calculate_glcm(img, msk, [1, 2], Ng, f2d, 0)
            wg, _ = checker.calculate_glcm(img, msk, [1, 2], Ng, f2d, 0)
            assert np.array_equal(g, wg)


def test_irregular_levels_fall_through_to_the_exact_kernels(checker):
    """a level outside 1..Ng under the mask: the pack flags it, the exact kernels say what the reference says (IndexError)"""
    from pyradiomics_amd import cmatrices as cm, _lib
    img, msk = _case((10, 12, 14), 300, 9)
    img[3, 4, 5] = 301
    msk[3, 4, 5] = True
    for fn, args in ((cm.calculate_glcm, ([1, 2], 300, False, 0)), (cm.calculate_glrlm, (300, 14, False, 0))):
        with pytest.raises(IndexError):
            fn(img, msk, *args)
        assert _lib.last_path() == "generic"
        with pytest.raises(IndexError):
            getattr(checker, fn.__name__)(img, msk, *args)
    img[3, 4, 5] = 0
    msk[3, 4, 5] = False                                               # outside the ROI anything goes
    g, _ = cm.calculate_glcm(img, msk, [1, 2], 300, False, 0)
    assert _lib.last_path() == "pairs" and np.array_equal(g, checker.calculate_glcm(img, msk, [1, 2], 300, False, 0)[0])


@pytest.mark.parametrize("shape,Ng,dist,alpha", [((20, 33, 40), 300, [1, 2], 0), ((20, 33, 40), 300, [1], 2), ((12, 40, 36), 1000, [1], 0),
                                                  ((9, 30, 31), 255, [1, 2], 1), ((40, 41), 500, [1, 2], 0)])
def test_gldm_ngtdm_beyond_the_byte_kernels_take_the_pairs_tier(shape, Ng, dist, alpha, checker):
    """GLDM (cmatrices.c:660-754) and NGTDM (:543-658) above 255 levels or with bin tables beyond kernels_neigh.h's LDS budget:
    counts bit for bit, NGTDM's float column within 1e-12 of the reference's raster-order sum"""
    from pyradiomics_amd import cmatrices as cm, _lib
    img, msk = _case(shape, Ng, 7, smooth=True, hole=len(shape) == 3)
    got = cm.calculate_gldm(img, msk, dist, Ng, alpha, False, 0)
    assert _lib.last_path() == "pairs", _lib.last_path()
    assert np.array_equal(got, checker.calculate_gldm(img, msk, dist, Ng, alpha, False, 0))
    a = cm.calculate_ngtdm(img, msk, dist, Ng, False, 0)
    assert _lib.last_path() == "pairs", _lib.last_path()
    b = checker.calculate_ngtdm(img, msk, dist, Ng, False, 0)
    assert np.array_equal(a[..., 0], b[..., 0]) and np.array_equal(a[..., 2], b[..., 2])
    np.testing.assert_allclose(a[..., 1], b[..., 1], rtol=1e-12, atol=1e-12)
    bad = img.copy()
    bad[tuple(np.argwhere(msk)[0])] = Ng + 1
    for fn, args in ((cm.calculate_gldm, (dist, Ng, alpha, False, 0)), (cm.calculate_ngtdm, (dist, Ng, False, 0))):
        with pytest.raises(IndexError):
            fn(bad, msk, *args)
        assert _lib.last_path() == "generic"


@pytest.mark.parametrize("shape,Ng,dist,f2d", [((20, 33, 40), 32, [1, 2], None), ((14, 21, 30), 32, [1, 2, 3], None), ((9, 30, 31), 200, [1, 2, 3], None),
                                              ((12, 40, 36), 64, [1, 2], 0), ((10, 25, 27), 500, [1, 2], 2), ((40, 41), 100, [1, 2, 3], None),
                                              ((5, 6, 7), 12, [1, 2], None), ((16, 20, 24), 32, [2], None)])
def test_ngtdm_over_full_boxes_of_neighbours_is_box_sums(shape, Ng, dist, f2d, checker):
    """NGTDM (cmatrices.c:543-658) for distances [1, 2] / [1, 2, 3] -- every offset of a 5^3 / 7^3 cube, 124 / 342 neighbours --
    as separable box sums (round 6), at any level count, with holes in the mask, force2D, volumes smaller than the box; a
    neighbour set that is NOT a full box (distances [2]: a shell) keeps its kernel.  Counts bit for bit, the float column
    within 1e-12 of the reference's raster-order sum; an irregular level raises what the reference raises"""
    from pyradiomics_amd import cmatrices as cm, _lib
    img, msk = _case(shape, Ng, 11, smooth=True, hole=len(shape) == 3)
    force, dim = f2d is not None, (f2d or 0)
    a = cm.calculate_ngtdm(img, msk, dist, Ng, force, dim)
    nd = len(shape) - (1 if force else 0)
    full_box = dist != [2] and (2 * max(dist) + 1) ** nd - 1 > 26      # (up to 26 neighbours the byte kernel keeps the call)
    assert _lib.last_path() == ("pairs" if full_box or Ng > 255 else "neigh"), _lib.last_path()
    b = checker.calculate_ngtdm(img, msk, dist, Ng, force, dim)
    assert np.array_equal(a[..., 0], b[..., 0]) and np.array_equal(a[..., 2], b[..., 2])
    np.testing.assert_allclose(a[..., 1], b[..., 1], rtol=1e-12, atol=1e-12)
    bad = img.copy()
    bad[tuple(np.argwhere(msk)[0])] = Ng + 1
    with pytest.raises(IndexError):
        cm.calculate_ngtdm(bad, msk, dist, Ng, force, dim)


def test_pairs_tier_at_config_size(checker):
    """VERDICT r4 item 6's targets, on the volume they are quoted for: 256^3, GLCM with distances [1, 2] at 32 levels, and
    GLCM + GLRLM at 255 levels, bit-exact; device ms printed (bench.py modes.fallback reports them)"""
    import torch
    from bench import make_volume
    from pyradiomics_amd import engine
    n = 256
    img_d, msk_d = make_volume(n, 32, "uniform", 1, torch.device("cuda", 0))
    g = engine.glcm(img_d, msk_d, 32, (1, 2))
    assert engine.last_path() == "pairs"
    ms = engine.last_device_ms()
    want, _ = checker.calculate_glcm(img_d.cpu().numpy(), msk_d.cpu().numpy().astype(bool), [1, 2], 32, False, 0)
    got = g[0].cpu().numpy()
    assert np.array_equal(got, np.asarray(want).reshape(got.shape))          # (the operator module's result carries the Nvox axis)
    print("GLCM distances [1,2], 32 levels, 256^3: %.3f ms" % ms)
    img255, msk255 = make_volume(n, 255, "smooth", 2, torch.device("cuda", 0))
    gg, rr, _ = engine.glcm_glrlm(img255, msk255, 255, n)
    assert engine.last_path() == "pairs"
    ms2 = engine.last_device_ms()
    wg, wr, _, _ = checker.glcm_glrlm_angle_sharded(img255.cpu().numpy(), msk255.cpu().numpy().astype(bool), 255, n)
    assert np.array_equal(gg.cpu().numpy(), wg) and np.array_equal(rr.cpu().numpy(), wr)
    print("GLCM + GLRLM, 255 levels, 256^3 smooth: %.3f ms" % ms2)
