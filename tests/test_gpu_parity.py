"""GPU parity: the HIP path (through the C ABI / pyradiomics_amd.cmatrices) against the CPU oracle on the same
seeded inputs.  Integer matrices must be bit-exact; the NGTDM float column within 1e-12 relative in segment
mode (exact-integer formulation, see include/pyradiomics_amd.h) and bit-exact in voxel mode."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cm():
    from pyradiomics_amd import cmatrices
    return cmatrices


def _vol(seed, shape, Ng, frac, smooth=False):
    rng = np.random.default_rng(seed)
    if smooth:
        from scipy import ndimage
        f = ndimage.gaussian_filter(rng.standard_normal(shape), 2.0)
        f = (f - f.min()) / (f.max() - f.min() + 1e-12)
        img = np.minimum((f * Ng).astype(np.int32) + 1, Ng)
    else:
        img = rng.integers(1, Ng + 1, size=shape).astype(np.int32)
    mask = np.ones(shape, bool) if frac >= 1.0 else rng.random(shape) < frac
    return img, mask


def _check_all(cm, oracle, img, mask, Ng, force2D=False, f2d=0, dist=(1,), alpha=0, vox=None, radius=0,
               expect_path=None):
    from pyradiomics_amd import _lib
    kw = {} if vox is None else dict(kernelRadius=radius, voxels=vox)
    dist = list(dist)
    Nr = max(img.shape)
    a, ang = cm.calculate_glcm(img, mask, dist, Ng, force2D, f2d, **kw)
    if expect_path:
        assert _lib.last_path() == expect_path
    b, bng = oracle.calculate_glcm(img, mask, dist, Ng, force2D, f2d, **kw)
    assert np.array_equal(ang, bng)
    assert a.dtype == np.float64 and a.shape == b.shape and np.array_equal(a, b), "GLCM"
    a, ang = cm.calculate_glrlm(img, mask, Ng, Nr, force2D, f2d, **kw)
    if expect_path:
        assert _lib.last_path() == expect_path
    b, bng = oracle.calculate_glrlm(img, mask, Ng, Nr, force2D, f2d, **kw)
    assert np.array_equal(ang, bng) and a.shape == b.shape and np.array_equal(a, b), "GLRLM"
    if vox is None:
        g, r, _ = cm.calculate_glcm_glrlm(img, mask, Ng, Nr, force2D, f2d)
        assert np.array_equal(r, b), "fused GLRLM"
        assert np.array_equal(g, oracle.calculate_glcm(img, mask, [1], Ng, force2D, f2d)[0]), "fused GLCM"
    a = cm.calculate_gldm(img, mask, dist, Ng, alpha, force2D, f2d, **kw)
    b = oracle.calculate_gldm(img, mask, dist, Ng, alpha, force2D, f2d, **kw)
    assert a.shape == b.shape and np.array_equal(a, b), "GLDM"
    a = cm.calculate_ngtdm(img, mask, dist, Ng, force2D, f2d, **kw)
    b = oracle.calculate_ngtdm(img, mask, dist, Ng, force2D, f2d, **kw)
    assert a.shape == b.shape
    assert np.array_equal(a[..., 0], b[..., 0]) and np.array_equal(a[..., 2], b[..., 2]), "NGTDM counts"
    if vox is None:
        np.testing.assert_allclose(a[..., 1], b[..., 1], rtol=1e-12, atol=0)
    else:
        assert np.array_equal(a[..., 1], b[..., 1]), "NGTDM voxel mode must be bit-exact"
    Ns = int(mask.sum())
    try:
        b = oracle.calculate_glszm(img, mask, Ng, Ns, force2D, f2d, **kw)
    except IndexError:   # e.g. empty mask: Ns = 0 exhausts the reference's scratch (cmatrices.c:274)
        with pytest.raises(IndexError):
            cm.calculate_glszm(img, mask, Ng, Ns, force2D, f2d, **kw)
    else:
        a = cm.calculate_glszm(img, mask, Ng, Ns, force2D, f2d, **kw)
        assert a.shape == b.shape and np.array_equal(a, b), "GLSZM"


SHAPES = [(5, 6, 7), (1, 9, 9), (9, 1, 5), (4, 4, 1), (12, 10, 8), (3, 70, 65), (33, 17, 130), (16, 16, 16)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("Ng", [3, 32])
@pytest.mark.parametrize("frac", [1.0, 0.6])
def test_segment_3d(cm, checker, shape, Ng, frac):
    img, mask = _vol(hash((shape, Ng)) % 1000, shape, Ng, frac)
    _check_all(cm, checker, img, mask, Ng, expect_path="sweep")


@pytest.mark.parametrize("shape", [(12, 10, 8), (3, 70, 65)])
@pytest.mark.parametrize("f2d", [0, 1, 2])
def test_segment_force2d(cm, checker, shape, f2d):
    img, mask = _vol(7, shape, 8, 0.7)
    _check_all(cm, checker, img, mask, 8, True, f2d, expect_path="sweep")


@pytest.mark.parametrize("shape", [(40, 50), (1, 64), (7,), (2, 3, 4, 3)])
def test_other_ranks(cm, checker, shape):
    img, mask = _vol(3, shape, 5, 0.8)
    _check_all(cm, checker, img, mask, 5)


def test_smooth_volume_long_runs(cm, checker):
    img, mask = _vol(11, (40, 48, 150), 16, 1.0, smooth=True)
    _check_all(cm, checker, img, mask, 16, expect_path="sweep")
    img[:] = 4   # one level everywhere: runs as long as the volume, GLSZM is a single zone
    _check_all(cm, checker, img, mask, 16, expect_path="sweep")


def test_distances_two(cm, checker):
    img, mask = _vol(5, (9, 10, 11), 6, 0.8)
    _check_all(cm, checker, img, mask, 6, dist=(1, 2), alpha=1)
    _check_all(cm, checker, img, mask, 6, dist=(2,), alpha=2)


def test_large_ng_generic(cm, checker):
    img, mask = _vol(6, (8, 9, 10), 300, 0.9)
    _check_all(cm, checker, img, mask, 300)


@pytest.mark.parametrize("force2D", [False, True])
def test_voxel_mode(cm, checker, force2D):
    img, mask = _vol(9, (10, 14, 12), 7, 0.7)
    rng = np.random.default_rng(0)
    co = np.array(np.where(mask))
    sel = rng.choice(co.shape[1], 200, replace=False)
    _check_all(cm, checker, img, mask, 7, force2D, 0, vox=co[:, sel], radius=2, expect_path=None)
    _check_all(cm, checker, img, mask, 7, force2D, 0, vox=co[:, sel[:5]], radius=1)


def test_irregular_levels_match_reference_semantics(cm, checker):
    """Levels <= 0 or > Ng under the mask: IndexError where the reference raises, aliased bins where it doesn't."""
    img, mask = _vol(2, (6, 7, 8), 5, 1.0)
    bad = img.copy()
    bad[2, 3, 4] = 0
    for fn, args in (("calculate_glcm", ([1], 5, False, 0)), ("calculate_glrlm", (5, 8, False, 0)),
                     ("calculate_gldm", ([1], 5, 0, False, 0)), ("calculate_ngtdm", ([1], 5, False, 0))):
        with pytest.raises(IndexError):
            getattr(checker, fn)(bad, mask, *args)
        with pytest.raises(IndexError):
            getattr(cm, fn)(bad, mask, *args)
    big = img.copy()
    big[1, 1, 1] = 6      # one level above Ng: the reference aliases / raises depending on the flat index
    for fn, args in (("calculate_glcm", ([1], 5, False, 0)), ("calculate_glrlm", (5, 8, False, 0))):
        try:
            want = getattr(checker, fn)(big, mask, *args)[0]
        except IndexError:
            with pytest.raises(IndexError):
                getattr(cm, fn)(big, mask, *args)
        else:
            assert np.array_equal(getattr(cm, fn)(big, mask, *args)[0], want)


def test_empty_mask_and_single_voxel(cm, checker):
    img, _ = _vol(1, (5, 5, 5), 4, 1.0)
    none = np.zeros(img.shape, bool)
    one = none.copy()
    one[2, 2, 2] = True
    for m in (none, one):
        _check_all(cm, checker, img, m, 4)


def test_argument_errors(cm):
    img, mask = _vol(1, (5, 5, 5), 4, 1.0)
    with pytest.raises(ValueError):
        cm.calculate_glcm(img, mask[0], [1], 4, False, 0)
    with pytest.raises(ValueError):
        cm.calculate_glcm(img, mask[:4], [1], 4, False, 0)
    with pytest.raises(RuntimeError):
        cm.calculate_glcm(img, mask, [0], 4, False, 0)
    with pytest.raises(RuntimeError):
        cm.calculate_glcm(img, mask, [1], 4, False, 0, 0, np.zeros((3, 2), int))
    with pytest.raises(RuntimeError):
        cm.calculate_glcm(img, mask, [1], 4, False, 0, 1, np.zeros((2, 2), int))


def test_glszm_zone_list_order(cm, checker):
    """tempData parity: zones listed in raster order of their first voxel (cmatrices.c:255-258)."""
    import ctypes as C
    from pyradiomics_amd import _lib
    img, mask = _vol(4, (9, 12, 20), 4, 0.8)
    cm.calculate_glszm(img, mask, 4, int(mask.sum()), False, 0)
    lib = _lib.load()
    cap = int(mask.sum())
    buf = np.empty(2 * cap + 1, dtype=np.intc)
    n = lib.prad_glszm_zones(0, buf.ctypes.data_as(C.POINTER(C.c_int)), cap)
    assert n > 0 and buf[2 * n] == -1
    # reference order from the oracle's calculate_glszm
    L = checker.L
    m2 = mask.copy()
    size = np.array(img.shape, dtype=np.intc)
    strides = np.array([s // 4 for s in img.strides], dtype=np.intc)
    bb = np.concatenate([np.zeros(3, np.intc), size - 1]).astype(np.intc)
    ang = checker.generate_angles(size, [1], 1, 0, 0)
    temp = np.empty(2 * cap + 1, dtype=np.intc)
    ip = C.POINTER(C.c_int)
    L.calculate_glszm(img.ctypes.data_as(ip), m2.ctypes.data_as(C.c_char_p), size.ctypes.data_as(ip),
                      bb.ctypes.data_as(ip), strides.ctypes.data_as(ip), ang.ctypes.data_as(ip), len(ang), 3,
                      temp.ctypes.data_as(ip), 4, cap, 1)
    assert np.array_equal(buf[:2 * n + 1], temp[:2 * n + 1])


def test_medium_volume_vs_oracle(cm, checker):
    """128^3 at 32 levels: large enough to exercise the persistent-grid paths, seconds for the oracle."""
    img, mask = _vol(0, (128, 128, 128), 32, 1.0)
    g, r, _ = cm.calculate_glcm_glrlm(img, mask, 32, 128, False, 0)
    assert np.array_equal(g, checker.calculate_glcm(img, mask, [1], 32, False, 0)[0])
    assert np.array_equal(r, checker.calculate_glrlm(img, mask, 32, 128, False, 0)[0])


@pytest.mark.parametrize("shape", [(9, 12, 16), (1, 20, 64), (6, 1, 8), (20, 24, 132), (4, 4, 4)])
@pytest.mark.parametrize("frac", [1.0, 0.55])
def test_gldm_ngtdm_packed_byte_path(cm, checker, shape, frac):
    """Nx % 4 == 0 takes the 4-voxels-per-lane packed-byte kernels (neigh4_kernel); all force2D variants, levels up
    to 255, alpha = 0 (packed) and alpha = 2 (per-neighbour kernel)"""
    for Ng in (5, 255):
        img, mask = _vol(sum(shape) + Ng, shape, Ng, frac)
        for force2D, f2d in ((False, 0), (True, 0), (True, 1), (True, 2)):
            try:
                want = checker.calculate_gldm(img, mask, [1], Ng, 0, force2D, f2d)
            except RuntimeError:
                continue        # no angle left for this shape / force2D combination
            assert np.array_equal(cm.calculate_gldm(img, mask, [1], Ng, 0, force2D, f2d), want)
            assert np.array_equal(cm.calculate_gldm(img, mask, [1], Ng, 2, force2D, f2d),
                                  checker.calculate_gldm(img, mask, [1], Ng, 2, force2D, f2d))
            a = cm.calculate_ngtdm(img, mask, [1], Ng, force2D, f2d)
            b = checker.calculate_ngtdm(img, mask, [1], Ng, force2D, f2d)
            assert np.array_equal(a[..., 0], b[..., 0]) and np.array_equal(a[..., 2], b[..., 2])
            np.testing.assert_allclose(a[..., 1], b[..., 1], rtol=1e-12, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,smooth", [((24, 40, 33), False), ((48, 64, 80), True), ((1, 50, 70), True)])
def test_glszm_compact_equals_dense(shape, smooth):
    """prad_glszm_sizes + prad_fill_glszm_compact_dev == the non-empty columns of the reference-layout matrix"""
    import torch
    from pyradiomics_amd import engine, cmatrices
    rng = np.random.default_rng(11)
    if smooth:
        import scipy.ndimage as ndi
        f = ndi.gaussian_filter(rng.standard_normal(shape), 2.0)
        img = (1 + np.floor((f - f.min()) / (np.ptp(f) + 1e-9) * 6)).astype(np.int32)
    else:
        img = rng.integers(1, 6, shape).astype(np.int32)
    mask = rng.random(shape) < 0.9
    dense = cmatrices.calculate_glszm(img, mask, 8, int(mask.sum()), False, 0)[0]
    P, sizes = engine.glszm_compact(torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda(), 8, int(mask.sum()))
    cols = np.flatnonzero(dense.sum(0))
    assert np.array_equal(sizes, cols + 1)
    assert np.array_equal(P.cpu().numpy(), dense[:, cols])
    P2, sizes2 = cmatrices.calculate_glszm_compact(img, mask, 8, int(mask.sum()), False, 0)
    assert np.array_equal(P2[0], dense[:, cols]) and np.array_equal(sizes2, sizes)


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["worklist", "worklist_full"])
def test_glszm_border_routes_agree_with_the_reference(cm, checker, route, monkeypatch):
    """Cross-tile zone pairs: the work list of the tile kernel (default) and the scan of the tile faces that takes over
    when the list is too small (PRAD_GLSZM_WORKCAP) -- both against cmatrices.c:94-297, 26- and 8-neighbourhoods, row
    lengths with and without whole quads, several tiles per axis"""
    import scipy.ndimage as ndi
    if route == "worklist_full":
        monkeypatch.setenv("PRAD_GLSZM_WORKCAP", "37")
    rng = np.random.default_rng(101)
    for shape, sigma, ng in (((21, 30, 136), 1.5, 5), ((17, 19, 131), 0.0, 3), ((9, 70, 200), 2.5, 4), ((3, 9, 260), 1.0, 2)):
        f = rng.standard_normal(shape)
        if sigma:
            f = ndi.gaussian_filter(f, sigma)
        img = (1 + np.floor((f - f.min()) / (np.ptp(f) + 1e-9) * ng * 0.999)).astype(np.int32)
        mask = rng.random(shape) < 0.93
        Ns = int(mask.sum())
        for force2D, f2d in ((False, 0), (True, 0)):
            want = checker.calculate_glszm(img, mask, ng, Ns, force2D, f2d)
            assert np.array_equal(cm.calculate_glszm(img, mask, ng, Ns, force2D, f2d), want), (shape, force2D)


@pytest.mark.gpu
def test_level_counts_and_tensor_inputs_of_the_operator_module():
    import torch
    from pyradiomics_amd import engine, cmatrices
    rng = np.random.default_rng(5)
    shape = (20, 31, 45)
    img = rng.integers(0, 12, shape).astype(np.int32)          # level 0 and levels > Ng under the mask -> counts[0]
    mask = rng.random(shape) < 0.7
    ti, tm = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()
    counts = engine.level_counts(ti, tm, 9)
    want = np.bincount(np.where((img >= 1) & (img <= 9), img, 0)[mask], minlength=10)
    assert np.array_equal(counts, want)
    img = np.maximum(img, 1)
    ti = torch.from_numpy(img).cuda()
    for d in ([1], [1, 2]):
        a, ang_a = cmatrices.calculate_glcm(ti, tm, np.array(d), 11, False, 0)
        b, ang_b = cmatrices.calculate_glcm(img, mask, np.array(d), 11, False, 0)
        assert np.array_equal(a, b) and np.array_equal(ang_a, ang_b)
    assert np.array_equal(cmatrices.calculate_glrlm(ti, tm, 11, 45, True, 0)[0], cmatrices.calculate_glrlm(img, mask, 11, 45, True, 0)[0])
    assert np.array_equal(cmatrices.calculate_glrlm(ti, tm, 11, 60, False, 0)[0], cmatrices.calculate_glrlm(img, mask, 11, 60, False, 0)[0])
    assert np.array_equal(cmatrices.calculate_gldm(ti, tm, np.array([1]), 11, 1, False, 0), cmatrices.calculate_gldm(img, mask, np.array([1]), 11, 1, False, 0))
    assert np.array_equal(cmatrices.calculate_ngtdm(ti, tm, np.array([1]), 11, False, 0), cmatrices.calculate_ngtdm(img, mask, np.array([1]), 11, False, 0))
    assert np.array_equal(cmatrices.calculate_glszm(ti, tm, 11, int(mask.sum()), False, 0), cmatrices.calculate_glszm(img, mask, 11, int(mask.sum()), False, 0))


@pytest.mark.parametrize("shape", [(6, 40, 300), (40, 6, 200), (150, 130, 8)])
def test_long_runs_all_three_mechanisms(cm, checker, shape):
    """runs longer than the LDS table's run-length slots: lengths just above RS go to the LDS long-run table,
    full-row / full-column runs of flat regions to the wave-aggregated L2 atomics -- along every axis"""
    rng = np.random.default_rng(21)
    Ng = 32
    img = rng.integers(1, Ng + 1, size=shape).astype(np.int32)
    img[: shape[0] // 2] = 7                                  # a flat half: runs as long as the axes
    img[:, : shape[1] // 3, :] = np.where(rng.random((shape[0], shape[1] // 3, shape[2])) < 0.02, 3, 9)   # 20..100-voxel runs
    mask = np.ones(shape, bool)
    mask[rng.random(shape) < 0.001] = False
    from pyradiomics_amd import _lib
    Nr = max(shape)
    g, r, ang = cm.calculate_glcm_glrlm(img, mask, Ng, Nr, False, 0)
    assert _lib.last_path() == "sweep"
    assert np.array_equal(r, checker.calculate_glrlm(img, mask, Ng, Nr, False, 0)[0])
    assert np.array_equal(g, checker.calculate_glcm(img, mask, [1], Ng, False, 0)[0])
    assert r[0, :, 80:, :].sum() > 0 and r[0, :, 20:78, :].sum() > 0      # both long-run paths were exercised


def test_full_size_512_properties():
    """BASELINE's headline size (512^3, 32 levels) is beyond what the CPU oracle finishes in seconds: check the
    size-independent identities every exact result must satisfy, on device-resident inputs, for all five matrices."""
    import torch
    from pyradiomics_amd import engine
    dev = torch.device("cuda", 0)
    N, Ng = 512, 32
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    img = torch.randint(1, Ng + 1, (N, N, N), generator=g, device=dev, dtype=torch.int32)
    img[:, :, : N // 4] = torch.div(img[:, :, : N // 4], 8, rounding_mode="floor") + 1      # a quarter with long runs
    mask = torch.rand((N, N, N), generator=g, device=dev) < 0.9
    counts = torch.bincount(img[mask], minlength=Ng + 1)[1:].to(torch.float64)
    nroi = int(mask.sum())
    glcm, glrlm, ang = engine.glcm_glrlm(img, mask, Ng, N)
    assert engine.last_path() == "sweep"
    lens = torch.arange(1, N + 1, device=dev, dtype=torch.float64)
    for a in range(ang.shape[0]):
        dz, dy, dx = (int(v) for v in ang[a])
        def sl(d):
            return (slice(0, N - d), slice(d, N)) if d >= 0 else (slice(-d, N), slice(0, N + d))
        (z0, z1), (y0, y1), (x0, x1) = sl(dz), sl(dy), sl(dx)
        both = mask[z0, y0, x0] & mask[z1, y1, x1]
        assert int(glcm[:, :, a].sum()) == int(both.sum()), "every ordered ROI pair is counted once"
        same = both & (img[z0, y0, x0] == img[z1, y1, x1])
        assert int(torch.diagonal(glcm[:, :, a]).sum()) == int(same.sum())
        # runs tile the ROI, level by level; pairs inside runs are the GLCM diagonal
        assert torch.equal((glrlm[:, :, a] * lens[None, :]).sum(1), counts)
        assert torch.equal((glrlm[:, :, a] * (lens - 1)[None, :]).sum(1), torch.diagonal(glcm[:, :, a]))
    # an x-angle sees each z-slab independently: additivity over a split of the volume
    half = engine.glcm_glrlm(img[: N // 2].contiguous(), mask[: N // 2].contiguous(), Ng, N)
    rest = engine.glcm_glrlm(img[N // 2:].contiguous(), mask[N // 2:].contiguous(), Ng, N)
    xa = [a for a in range(ang.shape[0]) if tuple(ang[a]) == (0, 0, 1)][0]
    assert torch.equal(half[0][:, :, xa] + rest[0][:, :, xa], glcm[:, :, xa])
    assert torch.equal(half[1][:, :, xa] + rest[1][:, :, xa], glrlm[:, :, xa])
    gldm = engine.gldm(img, mask, Ng, 0)
    assert torch.equal(gldm.sum(1), counts) and float(gldm[:, 27:].sum()) == 0
    ngtdm = engine.ngtdm(img, mask, Ng)
    assert torch.equal(ngtdm[:, 0], counts) and torch.equal(ngtdm[:, 2], torch.arange(1, Ng + 1, device=dev, dtype=torch.float64))
    P, sizes = engine.glszm_compact(img, mask, Ng, nroi)
    assert torch.equal((P * torch.from_numpy(sizes).to(dev, torch.float64)[None, :]).sum(1), counts), "zones tile the ROI"
    dense = engine.glszm(img, mask, Ng, nroi)
    assert torch.equal(dense[:, torch.from_numpy(sizes.astype(np.int64) - 1).to(dev)], P) and float(dense.sum()) == float(P.sum())
    st = engine.firstorder_stats(img, mask)
    assert st["Np"] == nroi and st["Minimum"] == float(img[mask].min()) and st["Maximum"] == float(img[mask].max())
    assert st["Energy"] == float((img[mask].to(torch.float64) ** 2).sum())
    cum = torch.cumsum(counts, 0).cpu().numpy()                      # order statistics of integer data from the histogram
    lvl = lambda k: float(np.searchsorted(cum, k + 1) + 1)           # k-th smallest ROI value (0-based)
    assert st["Median"] == (lvl((nroi - 1) // 2) + lvl(nroi // 2)) / 2 and st["P10"] >= 1 and st["P90"] <= Ng


def test_workspace_query_and_release(cm, checker):
    from pyradiomics_amd import engine
    img, mask = _vol(3, (20, 24, 28), 8, 0.8)
    want = checker.calculate_glszm(img, mask, 8, int(mask.sum()), False, 0)
    assert np.array_equal(cm.calculate_glszm(img, mask, 8, int(mask.sum()), False, 0), want)
    assert engine.workspace_bytes() > 0
    engine.release_workspace()
    assert engine.workspace_bytes() == 0
    with pytest.raises(ValueError):            # phase 2 without phase 1: the zone list went with the workspace
        _ = cm._lib.raise_for(cm._lib.load().prad_fill_glszm(np.zeros(8).ctypes.data, 1, 8, 1), "fill")
    assert np.array_equal(cm.calculate_glszm(img, mask, 8, int(mask.sum()), False, 0), want)
    _check_all(cm, checker, img, mask, 8)
