"""GPU parity of the two-table walk's side-job pack (csrc/kernels_sweepfw2.h PackWave16, round 5): in the deferred pipeline
the pack of volume N (int32 levels + mask -> 16-bit level*4 elements + plain level bytes) rides in the walk launch of
volume N-1, as the fused-table walk's has since round 3.  Every result bit-equal to the CPU checker (the reference's own
cmatrices.c when oracle/_ref travelled with the repo): full / ball / random masks (the three conversion paths of a piece),
levels with bits beyond the low half outside the mask, an irregular level under the mask, volumes of mixed shapes and of
mixed kernels (32 levels = fused table, 64 = two tables) back to back, and the 512^3 x 64-level volume of bench.py's
`modes.levels64` at full size against the synchronous call."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _levels(seed, shape, Ng, kind):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.integers(1, Ng + 1, size=shape).astype(np.int32)
    if kind == "flat":
        return np.full(shape, 1 + seed % Ng, np.int32)
    from scipy import ndimage
    sigma = 2.0 if kind == "smooth" else 6.0        # "blobs": runs of tens of voxels
    f = ndimage.gaussian_filter(rng.standard_normal(shape), sigma)
    f = (f - f.min()) / (f.max() - f.min() + 1e-12)
    return np.minimum((f * Ng).astype(np.int32) + 1, Ng)


def _mask(seed, shape, kind):
    rng = np.random.default_rng(seed + 99)
    if kind == "full":
        return np.ones(shape, bool)
    if kind == "random":
        return rng.random(shape) < 0.7
    zz, yy, xx = np.meshgrid(*[np.linspace(-1, 1, n) for n in shape], indexing="ij")
    return (zz ** 2 + yy ** 2 + xx ** 2) < 0.8        # ball: empty corners, long outside stretches


@pytest.fixture(scope="module")
def cm():
    from pyradiomics_amd import cmatrices
    return cmatrices


def _want(checker, img, mask, Ng, Nr=512):
    eg, _ = checker.calculate_glcm(img, mask, [1], Ng, False, 0)
    er, _ = checker.calculate_glrlm(img, mask, Ng, Nr, False, 0)
    return eg[0], er[0]


@pytest.mark.parametrize("Ng", [64, 100, 160])
def test_fw2_pipeline_packs_inside_the_previous_walk(cm, checker, Ng):
    """same-shape 64+-level volumes back to back: only the first one packs in a launch of its own (the "pack" timing family),
    partial masks and smooth levels go through the side job, the last volume is walked by the flush"""
    import torch
    from pyradiomics_amd import engine, _lib
    shapes = [(24, 30, 512), (24, 30, 512), (24, 30, 512), (20, 26, 256), (20, 26, 256), (18, 22, 300), (24, 30, 512), (26, 26, 128)]
    kinds = ["uniform", "smooth", "blobs", "uniform", "smooth", "uniform", "flat", "uniform"]
    masks = ["full", "ball", "random", "random", "full", "full", "ball", "ball"]
    vols = [(_levels(170 + i + Ng, s, Ng, k), _mask(180 + i, s, m)) for i, (s, k, m) in enumerate(zip(shapes, kinds, masks))]
    dev = [(torch.from_numpy(i).cuda(), torch.from_numpy(m.astype(np.uint8)).cuda()) for i, m in vols]
    engine.set_deferred_mode(1)
    try:
        got = [engine.glcm_glrlm(i, m, Ng, 512, deferred=True) for i, m in dev]
        assert _lib.last_variant() == "fw2"
        engine.deferred_status()
        for n, ((img, mask), (g, r, _)) in enumerate(zip(vols, got)):
            eg, er = _want(checker, img, mask, Ng)
            assert np.array_equal(g.cpu().numpy(), eg), "GLCM of volume %d %s" % (n, shapes[n])
            assert np.array_equal(r.cpu().numpy(), er), "GLRLM of volume %d %s" % (n, shapes[n])
        one = engine.glcm_glrlm(dev[0][0], dev[0][1], Ng, 512)
        one = (one[0].clone(), one[1].clone())
        engine.timing_begin()
        engine.glcm_glrlm(dev[0][0], dev[0][1], Ng, 512, deferred=True)
        engine.deferred_status()
        t_one_pack = engine.timing_ms("pack")
        engine.timing_end()
        engine.timing_begin()
        four = [engine.glcm_glrlm(dev[0][0], dev[0][1], Ng, 512, deferred=True) for _ in range(4)]
        engine.deferred_status()
        t_pack, t_sweep = engine.timing_ms("pack"), engine.timing_ms("sweep")
        engine.timing_end()
        assert t_sweep > 0 and t_one_pack > 0
        assert t_pack < 1.5 * t_one_pack + 0.01, (t_pack, t_one_pack)     # 4 volumes, ONE standalone pack
        for g, r, _ in four:
            assert torch.equal(g, one[0]) and torch.equal(r, one[1])
    finally:
        engine.set_deferred_mode(-1)


def test_fw2_pipeline_mixed_kernels(cm, checker):
    """fused-table (32 levels) and two-table (64 levels) volumes alternate: the side job writes the layout of the launch it
    rides in, so a volume of the other kind packs in a launch of its own; ragged rows (no multiple of 16) too"""
    import torch
    from pyradiomics_amd import engine
    plan = [(32, (24, 30, 512)), (64, (24, 30, 512)), (64, (24, 30, 512)), (32, (24, 30, 512)), (64, (20, 26, 300)),
            (64, (20, 26, 296)), (64, (20, 26, 296)), (32, (20, 26, 256)), (32, (20, 26, 256)), (64, (20, 26, 256))]
    vols = [(Ng, _levels(200 + i, s, Ng, "smooth" if i % 3 == 0 else "uniform"), _mask(210 + i, s, "ball" if i % 2 else "full"))
            for i, (Ng, s) in enumerate(plan)]
    dev = [(Ng, torch.from_numpy(i).cuda(), torch.from_numpy(m.astype(np.uint8)).cuda()) for Ng, i, m in vols]
    engine.set_deferred_mode(1)
    try:
        for _ in range(2):
            got = [engine.glcm_glrlm(i, m, Ng, 512, deferred=True) for Ng, i, m in dev]
            engine.deferred_status()
            for n, ((Ng, img, mask), (g, r, _)) in enumerate(zip(vols, got)):
                eg, er = _want(checker, img, mask, Ng)
                assert np.array_equal(g.cpu().numpy(), eg), "GLCM of volume %d %s" % (n, plan[n])
                assert np.array_equal(r.cpu().numpy(), er), "GLRLM of volume %d %s" % (n, plan[n])
    finally:
        engine.set_deferred_mode(-1)


@pytest.mark.parametrize("odd", [0, 65, 300, -1, 70000, 1 << 16, (1 << 16) + 5, 32768 + 3])
def test_fw2_side_job_pack_handles_odd_levels(cm, checker, odd):
    """two voxels per dword with packed-half arithmetic; whatever is not the plain case takes the exact per-voxel form: levels
    with bits beyond the low half OUTSIDE the mask are ignored (cmatrices.c:61-64 tests the mask first), an irregular level
    UNDER the mask is reported by the deferred status and the synchronous route says what the reference says"""
    import torch
    from pyradiomics_amd import engine, _lib
    shape, Ng = (24, 30, 512), 64
    engine.set_deferred_mode(1)
    try:
        base = _levels(5, shape, Ng, "uniform")
        mask = _mask(6, shape, "random")
        junk = base.copy()
        rng = np.random.default_rng(7)
        out = ~mask.astype(bool)
        junk[out] = rng.choice(np.array([0, -5, 255, 256, 1 << 20, -(1 << 30), 65, 1 << 16, 65535, 32768], dtype=np.int32), size=int(out.sum()))
        vols = [(base, mask), (junk, mask), (base, mask)]
        dev = [(torch.from_numpy(i).cuda(), torch.from_numpy(m.astype(np.uint8)).cuda()) for i, m in vols]
        got = [engine.glcm_glrlm(i, m, Ng, 512, deferred=True) for i, m in dev]      # volumes 2 and 3 pack as side jobs
        assert _lib.last_variant() == "fw2"
        engine.deferred_status()
        eg, er = _want(checker, base, mask, Ng)
        for g, r, _ in got:                                     # junk outside the mask changes nothing
            assert np.array_equal(g.cpu().numpy(), eg) and np.array_equal(r.cpu().numpy(), er)
        bad = base.copy()
        full = np.ones(shape, np.uint8)
        bad[11, 13, 200] = odd                                  # one irregular voxel under the (full) mask: the fast path's test
        devbad, devfull = torch.from_numpy(bad).cuda(), torch.from_numpy(full).cuda()
        engine.glcm_glrlm(dev[0][0], devfull, Ng, 512, deferred=True)
        engine.glcm_glrlm(devbad, devfull, Ng, 512, deferred=True)                 # packed by the side job
        engine.glcm_glrlm(dev[0][0], devfull, Ng, 512, deferred=True)
        with pytest.raises(_lib.DeferredLevelsError):
            engine.deferred_status()
        engine.deferred_status()
        try:
            want = checker.calculate_glcm(bad, full.astype(bool), [1], Ng, False, 0)
        except IndexError:
            with pytest.raises(IndexError):
                cm.calculate_glcm(bad, full.astype(bool), [1], Ng, False, 0)
        else:                                                   # (the reference aliases some out-of-range levels silently)
            assert np.array_equal(cm.calculate_glcm(bad, full.astype(bool), [1], Ng, False, 0)[0], want[0])
    finally:
        engine.set_deferred_mode(-1)


def test_fw2_pipeline_equals_lanes_and_synchronous_512(cm, monkeypatch):
    """bench.py's modes.levels64 volume at full size (512^3, 64 levels, uniform and smooth): what the deferred pipeline leaves
    behind == the synchronous call (whose pack is the standalone kernel), bit for bit.  (The synchronous 64-level route is
    tied to the reference C by test_gpu_fw.py's fw2 cases and scripts/r05_stress.py.)"""
    import torch
    import bench
    from pyradiomics_amd import engine
    dev = torch.device("cuda:0")
    for dist in ("uniform", "smooth"):
        im, mk = bench.make_volume(512, 64, dist, seed=3, device=dev)
        g0, r0, _ = engine.glcm_glrlm(im, mk, 64, 512)
        g0, r0 = g0.clone(), r0.clone()
        assert engine.last_variant() == "fw2"
        engine.set_deferred_mode(1)
        try:
            got = [engine.glcm_glrlm(im, mk, 64, 512, deferred=True) for _ in range(3)]
            engine.deferred_join()
            engine.deferred_status()
        finally:
            engine.set_deferred_mode(-1)
        for g, r, _ in got:
            assert torch.equal(g, g0) and torch.equal(r, r0), dist
        del im, mk


# ---- the x angle of a two-table volume on the 16-bit levels (sweep_fw2_rows_kernel, round 5) ---------------------------------
def _check_x(cm, checker, img, mask, Ng):
    from pyradiomics_amd import _lib
    Nr = int(max(img.shape))
    g, r, ang = cm.calculate_glcm_glrlm(img, mask, Ng, Nr, False, 0)
    assert _lib.last_path() == "sweep" and _lib.last_variant() == "fw2"
    eg, eang = checker.calculate_glcm(img, mask, [1], Ng, False, 0)
    er, _ = checker.calculate_glrlm(img, mask, Ng, Nr, False, 0)
    assert np.array_equal(ang, eang)
    bad = np.argwhere(g != eg)
    assert bad.size == 0, "GLCM differs at %d entries, first (i, j, angle) = %s: %s vs %s; angle %s" % (
        len(bad), bad[0], g[tuple(bad[0])], eg[tuple(bad[0])], ang[bad[0][-1]])
    bad = np.argwhere(r != er)
    assert bad.size == 0, "GLRLM differs at %d entries, first (i, len-1, angle) = %s: %s vs %s; angle %s" % (
        len(bad), bad[0], r[tuple(bad[0])], er[tuple(bad[0])], ang[bad[0][-1]])


@pytest.mark.parametrize("Ng", [64, 100, 128, 160, 45])
@pytest.mark.parametrize("nx", [512, 300, 72])
def test_fw2_rows_runs_around_the_last_length_slot(cm, checker, Ng, nx):
    """runs along x of every length from 18 to 81 -- the table keeps 48 .. 63 length slots at these level counts (prad_api.hip
    plan_sweep; PRAD_FW2_ROWS_WAVES moves them), so lengths below, at and above the last slot -- starting at every offset of
    the 8-voxel blocks the plain path vouches for, ending inside the row, at the row's last voxel and at the ROI's edge; a
    run longer than the slots goes to the global table"""
    rng = np.random.default_rng(Ng + nx)
    shape = (5, 46, nx)
    img = rng.integers(1, Ng + 1, size=shape).astype(np.int32)
    mask = np.ones(shape, bool)
    row = 0
    for z in range(shape[0]):
        for y in range(shape[1]):
            L = 18 + row % 64
            x = row % 9
            lv = 1 + row % Ng
            while x + L <= nx:
                img[z, y, x:x + L] = lv
                lv = 1 + (lv + 6) % Ng
                x += L
            if row % 5 == 0:                      # the last run ends with the row
                img[z, y, max(0, nx - L):] = 1 + (lv + 3) % Ng
            if row % 7 == 0 and nx > L + 20:      # ... or at a voxel outside the ROI
                mask[z, y, 10 + L] = False
                img[z, y, 10:10 + L] = 2
            row += 1
    _check_x(cm, checker, img, mask, Ng)
    _check_x(cm, checker, img, _mask(3, shape, "random"), Ng)


@pytest.mark.parametrize("Ng", [64, 129, 160])
def test_fw2_rows_kernel_equals_the_8_bit_rows_kernel(cm, Ng, monkeypatch):
    """the new x-angle kernel against the one it replaces (PRAD_NO_FW2_ROWS=1: kernels_sweep.h on the 8-bit copy), on a volume
    the CPU checker would take minutes for; more than 4096 rows (tile rows 8 apart) and fewer"""
    import torch
    from pyradiomics_amd import engine
    for shape in ((96, 80, 512), (30, 40, 200)):
        for kind, mkind in (("smooth", "ball"), ("uniform", "full"), ("blobs", "random")):
            img = torch.from_numpy(_levels(31 + Ng, shape, Ng, kind)).cuda()
            mask = torch.from_numpy(_mask(9, shape, mkind).astype(np.uint8)).cuda()
            monkeypatch.delenv("PRAD_NO_FW2_ROWS", raising=False)
            g1, r1, _ = engine.glcm_glrlm(img, mask, Ng, 512)
            g1, r1 = g1.clone(), r1.clone()
            assert engine.last_variant() == "fw2"
            monkeypatch.setenv("PRAD_NO_FW2_ROWS", "1")
            g0, r0, _ = engine.glcm_glrlm(img, mask, Ng, 512)
            assert torch.equal(g0, g1) and torch.equal(r0, r1), (shape, kind, mkind)
    monkeypatch.delenv("PRAD_NO_FW2_ROWS", raising=False)
