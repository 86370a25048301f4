"""First-order statistics: the device reductions (prad_firstorder_dev, prad_voxel_firstorder_dev) against the numpy
restatement of radiomics/firstorder.py in oracle/firstorder_oracle.py, which tests/test_oracle.py::test_golden_features
pins to the reference's baseline_firstorder.csv.  Tolerances: order statistics (min / max / percentiles / median) must
be identical; sums differ only by summation order (numpy pairwise vs. block tree), bounded here by 1e-11 relative."""
import numpy as np
import pytest

from helpers import load_baseline_features, prepared_case

EXACT = ("Np", "Minimum", "Maximum", "P10", "P25", "Median", "P75", "P90")


def _volume(dtype, shape, seed, frac=0.6):
    rng = np.random.default_rng(seed)
    if np.issubdtype(dtype, np.integer):
        img = rng.integers(-900, 1500, shape).astype(dtype)
    else:
        img = (rng.standard_normal(shape) * 37.5 + 11).astype(dtype)
    return img, rng.random(shape) < frac


def test_oracle_kernel_offsets_follow_the_operator_angles(oracle_port):
    """firstorder.py:57-67 builds the kernel from cMatrices.generate_angles(bbsize, 1..r, bidirectional) + centre"""
    from oracle import firstorder_oracle
    for bb, r, f2d in (((5, 5, 5), 2, None), ((3, 5, 2), 2, None), ((5, 5, 5), 1, 0), ((1, 4, 5), 2, None)):
        want = oracle_port.generate_angles(np.array(bb), np.arange(1, r + 1), True, f2d is not None, f2d or 0)
        want = {tuple(a) for a in want} | {(0, 0, 0)}
        got = {tuple(a) for a in firstorder_oracle.kernel_offsets(bb, r, f2d is not None, f2d or 0)}
        assert got == want, (bb, r, f2d)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.int16, np.int32, np.float32, np.float64])
@pytest.mark.parametrize("shape,frac", [((9, 30, 41), 0.6), ((1, 1, 7), 1.0), ((64, 64, 65), 0.05)])
def test_segment_statistics_vs_oracle(dtype, shape, frac):
    from oracle import firstorder_oracle
    from pyradiomics_amd import cmatrices
    img, mask = _volume(dtype, shape, 3, frac)
    mask.flat[0] = True
    for shift in (0.0, 2000.0):
        got = cmatrices.firstorder_stats(img, mask, shift)
        want = firstorder_oracle.firstorder_stats(img, mask, shift)
        assert set(got) == set(want)
        for k in want:
            if k in EXACT:
                assert got[k] == want[k], (k, got[k], want[k])
            elif np.isnan(want[k]):
                assert np.isnan(got[k]), k
            else:
                assert abs(got[k] - want[k]) <= 1e-11 * max(abs(want[k]), 1e-300) + 1e-9 * (k in ("m3",)) * abs(want["m2"]) ** 1.5, (k, got[k], want[k])


def _compare_stats(got, want):
    assert set(got) == set(want)
    for k in want:
        if k in EXACT:
            assert got[k] == want[k], (k, got[k], want[k])
        elif np.isnan(want[k]):
            assert np.isnan(got[k]), k
        else:
            assert abs(got[k] - want[k]) <= 1e-11 * max(abs(want[k]), 1e-300) + 1e-9 * (k in ("m3",)) * abs(want["m2"]) ** 1.5, (k, got[k], want[k])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["float64", "float32", "int16", "ties51", "ties2", "outlier"])
def test_large_roi_order_statistics_by_selection(kind, monkeypatch):
    """ROIs above 2^20 voxels take the histogram-selection route (no full sort): order statistics must still be the
    exact elements numpy picks, with ties, with an outlier stretching the bin range, for every input dtype.  Integer
    images with a value range below 32768 take the exact-histogram route instead (one pass after the min / max
    reduction); both routes are checked on them."""
    from oracle import firstorder_oracle
    from pyradiomics_amd import _lib, cmatrices
    shape = (128, 128, 130)
    rng = np.random.default_rng(21)
    if kind in ("float64", "float32", "int16"):
        img, mask = _volume(np.dtype(kind).type, shape, 5, 0.62)
    elif kind == "ties51":
        img, mask = rng.integers(0, 51, shape).astype(np.int16), rng.random(shape) < 0.7
    elif kind == "ties2":
        img, mask = (rng.random(shape) < 0.3).astype(np.int32) * 9 - 4, np.ones(shape, bool)
    else:
        img, mask = rng.standard_normal(shape) + 50.0, rng.random(shape) < 0.8
        img[3, 4, 5] = 1e9                      # nearly everything falls into the first bin: it is gathered whole
        mask[3, 4, 5] = True
    integer = kind in ("int16", "ties51", "ties2")
    for shift in (0.0, 1000.0):
        got = cmatrices.firstorder_stats(img, mask, shift)
        assert _lib.last_path() == ("firstorder-exact" if integer else "firstorder-select")
        _compare_stats(got, firstorder_oracle.firstorder_stats(img, mask, shift))
    if integer:
        monkeypatch.setenv("PRAD_FO_NO_EXACT", "1")
        got = cmatrices.firstorder_stats(img, mask, 0.0)
        assert _lib.last_path() == "firstorder-select"
        _compare_stats(got, firstorder_oracle.firstorder_stats(img, mask, 0.0))


@pytest.mark.gpu
def test_large_roi_heavy_ties_and_fallback_to_the_sort(monkeypatch):
    from oracle import firstorder_oracle
    from pyradiomics_amd import _lib, cmatrices
    shape = (160, 160, 172)                                   # 4.4 M voxels
    rng = np.random.default_rng(4)
    mask = np.ones(shape, bool)
    # two distinct values: the selected bins exceed the gather budget but each holds ONE value -> no gather, no sort
    img = (rng.random(shape) < 0.5).astype(np.int16) * 100
    for env, path in ((None, "firstorder-exact"), ("1", "firstorder-select")):
        if env:
            monkeypatch.setenv("PRAD_FO_NO_EXACT", env)
        got = cmatrices.firstorder_stats(img, mask, 0.0)
        assert _lib.last_path() == path
        _compare_stats(got, firstorder_oracle.firstorder_stats(img, mask, 0.0))
        # a discretised image (32 levels), the case of the level volumes of the texture classes
        lev = rng.integers(1, 33, shape).astype(np.int32)
        got = cmatrices.firstorder_stats(lev, mask, 0.0)
        assert _lib.last_path() == path
        _compare_stats(got, firstorder_oracle.firstorder_stats(lev, mask, 0.0))
    monkeypatch.delenv("PRAD_FO_NO_EXACT")
    # integers spread over more than 32768 values: back to the selection route
    wide = rng.integers(-40000, 40000, shape).astype(np.int32)
    got = cmatrices.firstorder_stats(wide, mask, 0.0)
    assert _lib.last_path() == "firstorder-select"
    _compare_stats(got, firstorder_oracle.firstorder_stats(wide, mask, 0.0))
    # the widest range the exact route takes (32768 distinct values), partial mask
    edge = rng.integers(-20000, 12768, shape).astype(np.int16)
    edge[0, 0, 0], edge[0, 0, 1] = -20000, 12767
    pm = rng.random(shape) < 0.6
    pm[0, 0, :2] = True
    got = cmatrices.firstorder_stats(edge, pm, 7.0)
    assert _lib.last_path() == "firstorder-exact"
    _compare_stats(got, firstorder_oracle.firstorder_stats(edge, pm, 7.0))
    # an outlier stretches the bin range: nearly the whole ROI shares bin 0 with many distinct values -> full sort
    img = rng.standard_normal(shape) + 50.0
    img[1, 2, 3] = 1e9
    got = cmatrices.firstorder_stats(img, mask, 0.0)
    assert _lib.last_path() == "firstorder-sort"
    _compare_stats(got, firstorder_oracle.firstorder_stats(img, mask, 0.0))


@pytest.mark.gpu
def test_segment_flat_region_and_single_voxel():
    from pyradiomics_amd import firstorder
    img = np.full((4, 5, 6), 7, dtype=np.int16)
    mask = np.ones(img.shape, dtype=np.int32)
    vals = firstorder.RadiomicsFirstOrder(img, mask, binWidth=25).execute()
    assert float(vals["Skewness"]) == 0 and float(vals["Kurtosis"]) == 0 and float(vals["Variance"]) == 0
    assert float(vals["Entropy"]) == pytest.approx(0, abs=1e-12) and float(vals["Uniformity"]) == 1
    assert float(vals["Median"]) == 7 and float(vals["RobustMeanAbsoluteDeviation"]) == 0
    one = np.zeros(img.shape, dtype=np.int32)
    one[1, 2, 3] = 1
    vals = firstorder.RadiomicsFirstOrder(img, one, binWidth=25, voxelArrayShift=3).execute()
    assert float(vals["Energy"]) == 100 and float(vals["RootMeanSquared"]) == 10 and float(vals["Range"]) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("cfgname", ["brain1", "brain2_resegmentation", "breast1"])
def test_golden_firstorder_host_and_device_routes(cfgname):
    """both call routes of the class (device tensors / host arrays uploaded by the operator module) on the GPU"""
    from pyradiomics_amd import backend, cmatrices, firstorder
    backend.set(cmatrices)
    cfg = load_baseline_features()[cfgname]
    image, mask, settings = prepared_case(cfg)
    for route in (True, False):
        got = firstorder.RadiomicsFirstOrder(image, mask, deviceResident=route, **settings).execute()
        for name, ref in cfg["features"]["firstorder"].items():
            assert abs(float(got[name]) - ref) <= 1e-9 * abs(ref) + 1e-12, (route, name, float(got[name]), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,radius,force2D,masked", [((7, 12, 11), 1, False, True), ((6, 9, 10), 2, False, True),
                                                          ((5, 14, 13), 2, True, True), ((6, 8, 9), 1, False, False),
                                                          ((2, 9, 9), 2, False, True)])
@pytest.mark.parametrize("dtype", [np.int16, np.float64])
def test_voxel_mode_vs_oracle(shape, radius, force2D, masked, dtype, oracle_port):
    from pyradiomics_amd import backend, cmatrices, firstorder
    img, roi = _volume(dtype, shape, 8, 0.7)
    mask = roi.astype(np.int32)
    kw = dict(binWidth=40, voxelBased=True, kernelRadius=radius, force2D=force2D, force2Ddimension=0,
              maskedKernel=masked, voxelArrayShift=17, voxelBatch=97, initValue=np.nan)
    res = {}
    for name, be in (("gpu", cmatrices), ("cpu", oracle_port)):
        backend.set(be)
        try:
            fc = firstorder.RadiomicsFirstOrder(img, mask, **kw)
            fc.enableAllFeatures()
            fc.enableFeatureByName("StandardDeviation")
            res[name] = {k: v.array for k, v in fc.execute().items()}
        finally:
            backend.set(cmatrices)
    assert set(res["gpu"]) == set(res["cpu"]) and len(res["gpu"]) == 19
    for k, want in res["cpu"].items():
        got = res["gpu"][k]
        assert np.array_equal(np.isnan(got), np.isnan(want)), k
        ok = ~np.isnan(want)
        assert ok.sum() == roi.sum()
        if k in ("Minimum", "Maximum", "Median", "10Percentile", "90Percentile", "InterquartileRange", "Range"):
            assert np.array_equal(got[ok], want[ok]), k
        else:
            scale = np.maximum(np.abs(want[ok]), 1e-9 if k in ("Skewness",) else 1e-300)
            assert np.all(np.abs(got[ok] - want[ok]) <= 1e-9 * scale + 1e-12), (k, np.abs(got[ok] - want[ok]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["float64", "float32", "peaked", "ties", "outlier"])
def test_queue_route_equals_the_synchronous_statistics(kind):
    """prad_firstorder_queue_dev (the passes' scalars stay in device memory, glue kernels do the host's arithmetic) against
    prad_firstorder_dev: the same bits; an image the queue declines says so in its verdict word"""
    import torch
    from pyradiomics_amd import engine
    shape = (128, 128, 130)
    rng = np.random.default_rng(4)
    if kind in ("float64", "float32"):
        img, mask = _volume(np.dtype(kind).type, shape, 5, 0.62)
    elif kind == "ties":        # integer-valued doubles: every gathered bin is one value repeated 40 000 times
        img, mask = rng.integers(0, 51, shape).astype(np.float64), rng.random(shape) < 0.95
    elif kind == "peaked":      # a wavelet detail band: most voxels near zero, long tails
        img, mask = rng.laplace(size=shape) * 20.0, rng.random(shape) < 0.9
    else:
        img, mask = rng.standard_normal(shape) + 50.0, rng.random(shape) < 0.8
        img[3, 4, 5] = 1e9      # nearly everything in the first histogram bin: more than the queue gathers
        mask[3, 4, 5] = True
    I = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    M = torch.from_numpy(mask.astype(np.uint8)).cuda()
    m = int(mask.sum())
    for shift in (0.0, 1000.0):
        want = engine.firstorder_stats(I, M, shift)
        if kind == "outlier":
            with pytest.raises(NotImplementedError):
                engine.firstorder_stats_queue(I, M, m, shift)
            v = engine.firstorder_stats_queue(I, M, m, shift, deferred=True)
            engine.deferred_status()
            assert int(v[15]) & 8
            continue
        got = engine.firstorder_stats_queue(I, M, m, shift)
        assert got[15] == 0
        for k, f in enumerate(engine.FIRSTORDER_FIELDS):
            assert np.array_equal(got[k], want[f], equal_nan=True), (f, got[k], want[f])
        q = engine.firstorder_stats_queue(I, M, m, shift, deferred=True)
        engine.deferred_status()
        assert np.array_equal(q, got, equal_nan=True)
    # a wrong ROI count is a verdict, not a wrong answer; integer images and small ROIs are declined up front
    v = engine.firstorder_stats_queue(I, M, m - 1, 0.0, deferred=True)
    engine.deferred_status()
    assert int(v[15]) & 1
    with pytest.raises(NotImplementedError):
        engine.firstorder_stats_queue(I.to(torch.int16), M, m, 0.0)
    with pytest.raises(NotImplementedError):
        engine.firstorder_stats_queue(I[:8].contiguous(), M[:8].contiguous(), int(mask[:8].sum()), 0.0)
