"""Pins the CPU oracle (tests' checker) and the Python host layer:
  1. oracle/texture_oracle.c  ==  the reference's own cmatrices.c (oracle/_ref, built from /root/reference) on
     seeded random volumes, odd shapes, partial masks, 1-4 dimensions, voxel-mode boxes  -- bit for bit;
  2. oracle + pyradiomics_amd feature classes reproduce the reference's golden MATRICES
     (data/baseline/<case>_<class>.npy; reference tests/test_matrices.py:35-65, tolerance there 1e-3, here exact
     for counts and 1e-12 for the normalised GLCM);
  3. ... and the reference's golden FEATURE values (data/baseline/baseline_<class>.csv; reference
     tests/test_features.py, tolerance there 3 %, here 1e-6 relative -- BASELINE.json's bar for derived features).
No GPU involved."""
import numpy as np
import pytest

from helpers import CLASSES, FEATURE_CLASSES, feature_class, load_baseline_features, load_case, prepared_case

SHAPES = [(5, 6, 7), (1, 9, 9), (9, 1, 5), (4, 4, 1), (12, 10, 8), (3, 3), (7,), (2, 3, 4, 3), (1, 1, 6)]


def _same(fa, fb):
    """both raise the same exception type, or both return equal arrays"""
    try:
        b = fb()
    except (IndexError, RuntimeError, ValueError) as e:
        with pytest.raises(type(e)):
            fa()
        return
    a = fa()
    if isinstance(b, tuple):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    else:
        assert a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("frac", [1.0, 0.6])
def test_port_equals_reference(oracle_port, oracle_ref, shape, frac):
    P, R = oracle_port, oracle_ref
    rng = np.random.default_rng(len(shape) * 100 + int(frac * 10))
    Ng = 6
    img = rng.integers(1, Ng + 1, size=shape).astype(np.int32)
    m = rng.random(shape) < frac
    Ns, Nr = int(m.sum()), max(shape)
    variants = [dict(force2D=False, f2d=0, kw={})]
    if len(shape) == 3:
        variants += [dict(force2D=True, f2d=d, kw={}) for d in (0, 1, 2)]
        co = np.array(np.where(m))
        if co.shape[1] >= 10:
            sel = rng.choice(co.shape[1], 10, replace=False)
            variants += [dict(force2D=False, f2d=0, kw=dict(kernelRadius=2, voxels=co[:, sel])),
                         dict(force2D=True, f2d=0, kw=dict(kernelRadius=1, voxels=co[:, sel]))]
    for v in variants:
        f2, fd, kw = v["force2D"], v["f2d"], v["kw"]
        for dist in ([1], [1, 2]):
            _same(lambda: P.calculate_glcm(img, m, dist, Ng, f2, fd, **kw), lambda: R.calculate_glcm(img, m, dist, Ng, f2, fd, **kw))
            _same(lambda: P.calculate_gldm(img, m, dist, Ng, 1, f2, fd, **kw), lambda: R.calculate_gldm(img, m, dist, Ng, 1, f2, fd, **kw))
            _same(lambda: P.calculate_ngtdm(img, m, dist, Ng, f2, fd, **kw), lambda: R.calculate_ngtdm(img, m, dist, Ng, f2, fd, **kw))
        _same(lambda: P.calculate_glrlm(img, m, Ng, Nr, f2, fd, **kw), lambda: R.calculate_glrlm(img, m, Ng, Nr, f2, fd, **kw))
        _same(lambda: P.calculate_glszm(img, m, Ng, Ns, f2, fd, **kw), lambda: R.calculate_glszm(img, m, Ng, Ns, f2, fd, **kw))


def test_port_angles_equal_reference(oracle_port, oracle_ref):
    for bi in (0, 1):
        for size in [(5, 5, 5), (1, 5, 5), (2, 2, 2), (3, 1, 3), (4, 4), (9,), (2, 3, 2, 3)]:
            for dist in ([1], [2], [1, 2], [3], [0]):
                for f2 in ((False, 0), (True, 0), (True, len(size) - 1)):
                    _same(lambda: oracle_port.generate_angles(size, dist, bi, *f2),
                          lambda: oracle_ref.generate_angles(size, dist, bi, *f2))


def test_glcm_docstring_example(oracle_port):
    """the worked example of the reference's GLCM class docstring (glcm.py:21-42)"""
    I = np.array([[1, 2, 5, 2, 3], [3, 2, 1, 3, 1], [1, 3, 5, 5, 2], [1, 1, 1, 1, 2], [1, 2, 4, 3, 5]])
    P, ang = oracle_port.calculate_glcm(I, np.ones(I.shape, bool), [1], 5, False, 0)
    a = [tuple(x) for x in ang.tolist()].index((0, 1))
    sym = P[0, :, :, a] + P[0, :, :, a].T
    want = np.array([[6, 4, 3, 0, 0], [4, 0, 2, 1, 3], [3, 2, 0, 1, 2], [0, 1, 1, 0, 0], [0, 3, 2, 0, 2]])
    assert np.array_equal(sym, want)


@pytest.fixture
def oracle_backend(oracle_port):
    from pyradiomics_amd import backend
    old = backend._cmatrices
    backend.set(oracle_port)
    yield oracle_port
    backend.set(old)


@pytest.mark.parametrize("case", ["brain1", "brain2", "breast1"])
@pytest.mark.parametrize("cls", CLASSES)
def test_golden_matrices(oracle_backend, case, cls):
    image, mask, golden = load_case(case)
    fc = feature_class(cls)(image, mask, binWidth=25, distances=[1], gldm_a=0, force2D=False, label=1)
    fc._initCalculation()
    P = getattr(fc, "P_" + cls)
    want = golden[cls]
    assert P.shape[0] == 1 and P[0].shape == want.shape
    if cls == "glcm":
        np.testing.assert_allclose(P[0], want, rtol=0, atol=1e-12)
    else:
        assert np.array_equal(P[0], want), "max abs diff %g" % np.abs(P[0] - want).max()


@pytest.mark.parametrize("cfgname", sorted(load_baseline_features()))
def test_golden_features(oracle_backend, cfgname):
    cfg = load_baseline_features()[cfgname]
    image, mask, settings = prepared_case(cfg)
    for cls in FEATURE_CLASSES:
        if cls not in cfg["features"]:     # e.g. the weighting-norm config only exists for GLCM / GLRLM
            continue
        fc = feature_class(cls)(image, mask, **settings)
        got = fc.execute()
        want = cfg["features"][cls]
        assert set(got) == set(want), "feature-name surface of %s" % cls   # reference test_features.py:40-49
        for name, ref in want.items():
            val = float(got[name])
            if ref == 0 or not np.isfinite(ref):
                assert val == ref or (np.isnan(val) and np.isnan(ref)) or abs(val) < 1e-12, (cls, name, val, ref)
            else:
                assert abs(val - ref) <= 1e-6 * abs(ref), (cls, name, val, ref)
