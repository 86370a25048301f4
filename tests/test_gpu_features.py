"""GPU: BASELINE config 1 and friends through the product path -- feature classes on the HIP cMatrices backend --
against the reference's golden matrices (exact counts / 1e-12 normalised GLCM) and golden feature values (1e-6
relative), plus raw-matrix bit-exactness against the oracle on the real brain1 / brain2 / breast1 ROIs."""
import numpy as np
import pytest

from helpers import CLASSES, FEATURE_CLASSES, feature_class, load_baseline_features, load_case, prepared_case

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def hip_backend():
    from pyradiomics_amd import backend, cmatrices
    backend.set(cmatrices)
    yield
    backend.set(None)


@pytest.mark.parametrize("case", ["brain1", "brain2", "breast1"])
@pytest.mark.parametrize("cls", CLASSES)
def test_golden_matrices_on_gpu(case, cls):
    image, mask, golden = load_case(case)
    fc = feature_class(cls)(image, mask, binWidth=25, distances=[1], gldm_a=0, force2D=False, label=1)
    fc._initCalculation()
    P = getattr(fc, "P_" + cls)[0]
    if cls == "glcm":
        np.testing.assert_allclose(P, golden[cls], rtol=0, atol=1e-12)
    elif cls == "ngtdm":
        assert np.array_equal(P[:, 0], golden[cls][:, 0]) and np.array_equal(P[:, 2], golden[cls][:, 2])
        np.testing.assert_allclose(P[:, 1], golden[cls][:, 1], rtol=1e-12)
    else:
        assert np.array_equal(P, golden[cls])


@pytest.mark.parametrize("cfgname", sorted(load_baseline_features()))
def test_golden_features_on_gpu(cfgname):
    cfg = load_baseline_features()[cfgname]
    image, mask, settings = prepared_case(cfg)
    for cls in FEATURE_CLASSES:
        if cls not in cfg["features"]:
            continue
        got = feature_class(cls)(image, mask, **settings).execute()
        want = cfg["features"][cls]
        assert set(got) == set(want)
        for name, ref in want.items():
            val = float(got[name])
            if ref == 0:
                assert abs(val) < 1e-12, (cls, name, val)
            else:
                assert abs(val - ref) <= 1e-6 * abs(ref), (cls, name, val, ref)


@pytest.mark.parametrize("case", ["brain1", "brain2", "breast1"])
def test_raw_matrices_bit_exact_vs_oracle(case, checker):
    oracle_port = checker
    """the float64 arrays at the cMatrices boundary, before any numpy post-processing"""
    from pyradiomics_amd import cmatrices as cm, imageoperations
    image, mask, _ = load_case(case)
    m = mask.array == 1
    lv, _ = imageoperations.binImage(image.array, m, binWidth=25)
    Ng, Nr, Ns = int(lv[m].max()), max(lv.shape), int(m.sum())
    assert np.array_equal(cm.calculate_glcm(lv, m, [1], Ng, False, 0)[0], oracle_port.calculate_glcm(lv, m, [1], Ng, False, 0)[0])
    assert np.array_equal(cm.calculate_glrlm(lv, m, Ng, Nr, False, 0)[0], oracle_port.calculate_glrlm(lv, m, Ng, Nr, False, 0)[0])
    assert np.array_equal(cm.calculate_gldm(lv, m, [1], Ng, 0, False, 0), oracle_port.calculate_gldm(lv, m, [1], Ng, 0, False, 0))
    assert np.array_equal(cm.calculate_glszm(lv, m, Ng, Ns, False, 0), oracle_port.calculate_glszm(lv, m, Ng, Ns, False, 0))
    a, b = cm.calculate_ngtdm(lv, m, [1], Ng, False, 0), oracle_port.calculate_ngtdm(lv, m, [1], Ng, False, 0)
    assert np.array_equal(a[..., 0], b[..., 0])
    np.testing.assert_allclose(a[..., 1], b[..., 1], rtol=1e-12)


def test_voxel_based_glcm_map(oracle_port):
    """helloVoxel-style parameters (exampleVoxel.yaml: force2D, kernelRadius 2, maskedKernel, JointEntropy) on the
    brain2 ROI: the feature map computed on the GPU backend equals the one computed on the oracle backend"""
    from pyradiomics_amd import backend, cmatrices, glcm
    image, mask, _ = load_case("brain2")
    kw = dict(binWidth=25, force2D=True, force2Ddimension=0, kernelRadius=2, maskedKernel=True, initValue=np.nan,
              voxelBatch=200, voxelBased=True, label=1)
    maps = {}
    for name, be in (("gpu", cmatrices), ("oracle", oracle_port)):
        backend.set(be)
        fc = glcm.RadiomicsGLCM(image, mask, **kw)
        fc.enableFeatureByName("JointEntropy")
        fc.enableFeatureByName("Contrast")
        maps[name] = {k: v.array for k, v in fc.execute().items()}
    for k in maps["gpu"]:
        a, b = maps["gpu"][k], maps["oracle"][k]
        assert np.array_equal(np.isnan(a), np.isnan(b))
        np.testing.assert_allclose(a[~np.isnan(a)], b[~np.isnan(b)], rtol=1e-12, atol=0)


def _assert_degenerate(fc, coords, name, symmetrical):
    """the kernels at `coords` must be the reference's ill-conditioned special cases: for Correlation some angle whose
    row or column marginal is a single level (sigma = 0: glcm.py:409-410), for Imc2 some angle whose matrix is the outer
    product of its marginals (HXY2 == HXY: glcm.py:641-647)"""
    from pyradiomics_amd import cmatrices
    st = fc.settings
    host = lambda x: x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)
    P, _ = cmatrices.calculate_glcm(host(fc.imageArray), host(fc.maskArray), np.array(st.get("distances", [1])), fc.coefficients["Ng"],
                                    st.get("force2D", False), st.get("force2Ddimension", 0), st.get("kernelRadius", 1),
                                    coords)
    for k in range(P.shape[0]):
        found = False
        for a in range(P.shape[3]):
            M = P[k, :, :, a]
            if symmetrical:
                M = M + M.T
            if M.sum() == 0:
                continue
            M = M / M.sum()
            px, py = M.sum(1), M.sum(0)
            if name == "Correlation":
                found |= np.count_nonzero(px) == 1 or np.count_nonzero(py) == 1
            else:
                found |= np.allclose(M, np.outer(px, py), rtol=0, atol=1e-15)
        assert found, "%s: kernel at %s differs between the fused and the matrix route without being degenerate" % (
            name, coords[:, k])


@pytest.mark.parametrize("force2D", [True, False])
@pytest.mark.parametrize("symmetrical", [True, False])
def test_fused_voxel_glcm_equals_matrix_route(force2D, symmetrical):
    """every feature of the fused voxel kernel against the reference's route (per-kernel matrices + numpy formulas)
    on the same backend; kernels at the ROI border (empty angles, clamped windows) included"""
    from pyradiomics_amd import cmatrices, glcm
    image, mask, _ = load_case("brain2")
    names = [n for n in cmatrices.VOXEL_GLCM_FEATURES]
    kw = dict(binWidth=25, force2D=force2D, force2Ddimension=0, kernelRadius=2, maskedKernel=True, initValue=np.nan,
              voxelBased=True, label=1, symmetricalGLCM=symmetrical)
    maps = {}
    for fused in (True, False):
        fc = glcm.RadiomicsGLCM(image, mask, fusedVoxel=fused, **kw)
        for n in names:
            fc.enableFeatureByName(n)
        maps[fused] = {k: v.array for k, v in fc.execute().items()}
    for n in names:
        a, b = maps[True][n], maps[False][n]
        assert np.array_equal(np.isnan(a), np.isnan(b)), n
        ok = ~np.isnan(a)
        if n in ("Correlation", "Imc2"):
            # Ill-conditioned special cases of the reference.  Correlation: kernels in which one marginal is a single
            # level have sigma = 0 mathematically; numpy's ux carries rounding noise there, so its "sigma == 0 -> 1"
            # rule (glcm.py:409-410) fires erratically, while the fused kernel computes ux from integer sums and hits
            # it exactly.  Imc2: for rank-1 windows HXY2 == HXY mathematically and the reference's outcome (0, a
            # 1e-8 value, or NaN dropped by nanmean; glcm.py:641-647) is decided by the last bit of two log sums.
            # Allow those isolated kernels.
            # Allow those kernels -- and ONLY those: every differing kernel is enumerated and its per-kernel matrix
            # (from the operator itself) must show the degenerate structure.
            bad = ok & ~np.isclose(a, b, rtol=1e-9, atol=1e-12, equal_nan=True)
            assert bad.sum() < 0.01 * ok.sum(), "%s differs on %d kernels" % (n, bad.sum())
            if bad.any():
                _assert_degenerate(fc, np.array(np.nonzero(bad)), n, symmetrical)
            continue
        np.testing.assert_allclose(a[ok], b[ok], rtol=1e-9, atol=1e-12, err_msg=n)


def test_fused_voxel_glcm_unmasked_kernel_and_3d_radius1():
    from pyradiomics_amd import glcm
    image, mask, _ = load_case("breast1")
    kw = dict(binWidth=25, kernelRadius=1, maskedKernel=False, initValue=0, voxelBased=True, label=1)
    out = {}
    for fused in (True, False):
        fc = glcm.RadiomicsGLCM(image, mask, fusedVoxel=fused, **kw)
        for n in ("JointEntropy", "Idm", "Imc2", "Correlation"):
            fc.enableFeatureByName(n)
        out[fused] = {k: v.array for k, v in fc.execute().items()}
    for n in out[True]:
        if n in ("Correlation", "Imc2"):
            assert (~np.isclose(out[True][n], out[False][n], rtol=1e-9, atol=1e-12, equal_nan=True)).mean() < 0.01
            continue
        np.testing.assert_allclose(out[True][n], out[False][n], rtol=1e-9, atol=1e-12, equal_nan=True, err_msg=n)


@pytest.mark.parametrize("cls", ["glrlm", "glszm", "gldm", "ngtdm"])
@pytest.mark.parametrize("case,force2D,radius,masked", [("brain2", True, 2, True), ("brain2", False, 1, True),
                                                        ("breast1", False, 2, True), ("breast1", False, 1, False),
                                                        ("brain2", True, 3, False)])
def test_fused_voxel_texture_equals_matrix_route(cls, case, force2D, radius, masked):
    """fused voxel kernels of GLRLM / GLSZM / GLDM / NGTDM against the reference's route (per-kernel matrices from the
    generic kernels + numpy formulas): every non-deprecated feature, kernels at the ROI border (clamped windows, empty
    GLRLM angles) and unmasked kernels included"""
    from pyradiomics_amd import cmatrices
    image, mask, _ = load_case(case)
    kw = dict(binWidth=25, force2D=force2D, force2Ddimension=0, kernelRadius=radius, maskedKernel=masked,
              initValue=np.nan, voxelBased=True, label=1, voxelBatch=173)
    if cls == "gldm":
        kw["gldm_a"] = 1
    maps = {}
    for fused in (True, False):
        fc = feature_class(cls)(image, mask, fusedVoxel=fused, **kw)
        maps[fused] = {k: v.array for k, v in fc.execute().items()}
        if fused:
            assert cmatrices._lib.last_path() == "voxel-fused"
    assert set(maps[True]) == set(maps[False]) and len(maps[True]) >= 5
    for n in maps[True]:
        a, b = maps[True][n], maps[False][n]
        assert np.array_equal(np.isnan(a), np.isnan(b)), n
        ok = ~np.isnan(a)
        assert ok.sum() > 0
        np.testing.assert_allclose(a[ok], b[ok], rtol=1e-9, atol=1e-12, err_msg="%s %s" % (cls, n))


@pytest.mark.parametrize("cfgname", ["brain1", "brain2_2d", "breast1_combined", "brain1_FBN", "breast1_flatRegion"])
def test_segment_features_on_device_equal_numpy_formulas(cfgname):
    """fusedSegment (matrix and formulas on the GPU, prad_glcm_features_dev / prad_zone_matrix_features_dev) against the
    numpy formulas applied to the same device-built matrices, every non-deprecated feature, golden configurations"""
    cfg = load_baseline_features()[cfgname]
    image, mask, settings = prepared_case(cfg)
    for cls in ("glcm", "glrlm", "glszm", "gldm", "ngtdm"):
        if cls not in cfg["features"]:
            continue
        vals = {}
        for fused in (True, False):
            fc = feature_class(cls)(image, mask, fusedSegment=fused, **settings)
            vals[fused] = fc.execute()
        assert set(vals[True]) == set(vals[False]) == set(cfg["features"][cls])
        for n, b in vals[False].items():
            a, b = float(vals[True][n]), float(b)
            assert a == b or (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-10 * abs(b), (cls, n, a, b)


def test_segment_features_kernels_on_degenerate_matrices():
    import torch
    from pyradiomics_amd import engine
    dev = torch.device("cuda", 0)
    g = torch.zeros((5, 5, 3), dtype=torch.float64, device=dev)
    g[2, 2, 0] = 7                      # a single level: sigma = 0 -> Correlation 1, Imc1 0, Imc2 0 (glcm.py special cases)
    g[1, 3, 2] = 4
    f, empty = engine.glcm_features(g, True)
    assert list(empty) == [False, True, False] and np.isnan(f[1]).all()
    assert f[0, 6] == 1 and f[0, 12] == 0 and f[0, 13] == 0 and f[0, 11] == pytest.approx(0, abs=1e-12) and f[0, 19] == 1
    assert f[2, 5] == 4 and f[2, 1] == 3            # Contrast (2-4)^2, JointAverage of the symmetrised pair
    P = torch.zeros((4, 6), dtype=torch.float64, device=dev)
    z, e = engine.zone_matrix_features(P, np.arange(1, 7))
    assert e[0] and np.isnan(z[0]).all()
    P[1, 2] = 5
    z, e = engine.zone_matrix_features(P, np.arange(1, 7))
    assert not e[0] and z[0, 0] == pytest.approx(1 / 9) and z[0, 6] == pytest.approx(1 / 3) and z[0, 9] == pytest.approx(0, abs=1e-12)


def test_voxel_mcc_on_the_device():
    """MCC (an eigenvalue problem per kernel and angle, csrc/kernels_mcc.h) is evaluated on the device too: no per-kernel
    matrix reaches the host, and the map equals the reference's route (matrix + the numpy formula of glcm.py:665-707)"""
    from pyradiomics_amd import _lib, cmatrices, glcm
    for case, radius, f2d in (("breast1", 1, False), ("brain2", 2, True)):
        image, mask, _ = load_case(case)
        kw = dict(binWidth=25, kernelRadius=radius, maskedKernel=True, initValue=np.nan, voxelBased=True, label=1,
                  force2D=f2d, force2Ddimension=0)
        out = {}
        for fused in (True, False):
            fc = glcm.RadiomicsGLCM(image, mask, fusedVoxel=fused, **kw)
            for n in ("MCC", "JointEntropy", "Contrast"):
                fc.enableFeatureByName(n)
            calls, mats = [], []
            orig, orig_m = cmatrices.voxel_glcm_features, cmatrices.calculate_glcm
            cmatrices.voxel_glcm_features = lambda *a, **k: (calls.append(a[8]), orig(*a, **k))[1]
            cmatrices.calculate_glcm = lambda *a, **k: (mats.append(1), orig_m(*a, **k))[1]
            try:
                out[fused] = {k: v.array for k, v in fc.execute().items()}
                if fused:
                    assert _lib.last_path() == "voxel-fused"
            finally:
                cmatrices.voxel_glcm_features, cmatrices.calculate_glcm = orig, orig_m
            assert (calls == [["MCC", "JointEntropy", "Contrast"]]) == fused
            assert (len(mats) == 0) == fused          # the fused route never builds (Nvox, Ng, Ng, Na)
        for n in out[True]:
            np.testing.assert_allclose(out[True][n], out[False][n], rtol=1e-9, atol=1e-10, equal_nan=True, err_msg=n)


@pytest.mark.parametrize("Ng", [62, 63, 64])
def test_voxel_mcc_at_the_top_of_its_level_range(Ng):
    """Ng = 63 / 64: two per-wave scratch areas of the voxel MCC kernel exceed the 160 KiB of LDS a workgroup may
    declare (2 x 84 KB), the launch then runs one wave per workgroup instead of failing with PRAD_E_HIP; values against
    the matrix route + the numpy formula of glcm.py:665-707"""
    from pyradiomics_amd import _lib, glcm
    from pyradiomics_amd.image import Image
    rng = np.random.default_rng(Ng)
    arr = rng.integers(0, Ng, size=(6, 9, 10)).astype(np.int16)
    arr.flat[0], arr.flat[1] = 0, Ng - 1
    msk = np.ones(arr.shape, dtype=np.int32)
    kw = dict(binWidth=1, kernelRadius=1, maskedKernel=True, initValue=np.nan, voxelBased=True, label=1)
    out = {}
    for fused in (True, False):
        fc = glcm.RadiomicsGLCM(Image(arr), Image(msk), fusedVoxel=fused, **kw)
        fc.enableFeatureByName("MCC")
        out[fused] = fc.execute()["MCC"].array
        assert fc.coefficients["Ng"] == Ng
        if fused:
            assert _lib.last_path() == "voxel-fused"
    np.testing.assert_allclose(out[True], out[False], rtol=1e-9, atol=1e-10, equal_nan=True)


def test_segment_mcc_on_the_device_and_special_cases():
    import torch
    from pyradiomics_amd import engine
    rng = np.random.default_rng(4)
    for Ng, sym in ((2, True), (5, False), (32, True), (64, True)):
        P = rng.integers(0, 50, size=(Ng, Ng, 3)).astype(np.float64)
        P[:, :, 1] = 0                              # an angle without pairs -> NaN
        P[:, :, 2] = 0
        P[Ng // 2, Ng // 2, 2] = 7                  # one grey level only -> second eigenvalue 0
        got = engine.glcm_mcc(torch.from_numpy(P).cuda(), sym)
        M = P[:, :, 0] + (P[:, :, 0].T if sym else 0)
        p = M / M.sum()
        A = p / np.sqrt(np.outer(p.sum(1), p.sum(0)) + np.spacing(1))
        want = np.linalg.svd(A, compute_uv=False)[1]
        assert got[0] == pytest.approx(want, rel=1e-10) and np.isnan(got[1]) and got[2] == 0.0
    # the tridiagonal route (Householder + Sturm counts) on the shapes that strain it: odd sizes, levels that never
    # occur, a banded matrix (weak coupling: small second eigenvalue), a block matrix with lambda_2 = lambda_3 = ... = 1
    for Ng in (3, 7, 17, 33, 64):
        P = rng.integers(0, 30, size=(Ng, Ng, 4)).astype(np.float64)
        P[Ng // 3] = 0
        P[:, Ng // 3] = 0
        band = np.abs(np.subtract.outer(np.arange(Ng), np.arange(Ng))) <= 1
        P[:, :, 1] = band * rng.integers(1, 30, size=(Ng, Ng)) + 1e-3
        P[:, :, 2] = np.eye(Ng) * 5
        P[:, :, 3] = np.kron(np.eye((Ng + 1) // 2), np.ones((2, 2)))[:Ng, :Ng] * 3
        got = engine.glcm_mcc(torch.from_numpy(P).cuda(), True)
        for a in range(4):
            M = P[:, :, a] + P[:, :, a].T
            p = M / M.sum()
            A = p / np.sqrt(np.outer(p.sum(1), p.sum(0)) + np.spacing(1))
            keep = p.sum(1) > 0
            sv = np.linalg.svd(A[np.ix_(keep, keep)], compute_uv=False)
            want = sv[1] if len(sv) > 1 else 0.0
            assert got[a] == pytest.approx(want, rel=1e-9, abs=1e-12), (Ng, a, got[a], want)
    with pytest.raises(NotImplementedError):          # more than 64 grey levels occur: host route
        engine.glcm_mcc(torch.from_numpy(rng.integers(1, 9, size=(80, 80, 1)).astype(np.float64)).cuda(), True)


@pytest.mark.parametrize("force2D,radius,symmetrical", [(True, 2, True), (True, 3, True), (False, 1, True), (True, 2, False),
                                                       (False, 1, False)])
def test_light_voxel_glcm_kernel_equals_the_general_one(force2D, radius, symmetrical, monkeypatch):
    """the entropy / energy / maximum / average maps take a kernel of their own on windows of at most 64 voxels
    (division-free occurrence sums, tabulated log2): same maps as the general kernel within 1e-12, on a masked volume
    with holes (empty angles, partial windows at the border)"""
    import torch
    from pyradiomics_amd import engine
    monkeypatch.setenv("PRAD_VOX_NO_SLIDE", "1")     # (this test is about the two from-scratch window kernels)
    rng = np.random.default_rng(11)
    shape = (18, 40, 44)
    img = rng.integers(1, 17, shape).astype(np.int32)
    msk = (rng.random(shape) < 0.8).astype(np.uint8)
    msk[5:9, 10:20, 12:30] = 0                       # a hole larger than the window: centres with no pair at all
    di, dm = torch.from_numpy(img).cuda(), torch.from_numpy(msk).cuda()
    vox = torch.nonzero(dm).T.to(torch.int32).contiguous()
    feats = ["JointEntropy", "JointEnergy", "MaximumProbability", "JointAverage"]
    kw = dict(kernelRadius=radius, force2D=force2D, force2Ddimension=0, symmetrical=symmetrical)
    light = {k: v.cpu().numpy() for k, v in engine.voxel_glcm_features(di, dm, 16, vox, feats, **kw).items()}
    one = engine.voxel_glcm_features(di, dm, 16, vox, ["JointEntropy"], **kw)["JointEntropy"].cpu().numpy()
    monkeypatch.setenv("PRAD_VOX_NO_LIGHT", "1")
    gen = {k: v.cpu().numpy() for k, v in engine.voxel_glcm_features(di, dm, 16, vox, feats, **kw).items()}
    for k in feats:
        np.testing.assert_allclose(light[k], gen[k], rtol=1e-12, atol=1e-13, equal_nan=True, err_msg=k)
    np.testing.assert_array_equal(one, light["JointEntropy"])


@pytest.mark.gpu
def test_case_pipeline_enqueue_equals_class_by_class():
    """featureextractor.computeFeatures queues GLCM / GLRLM / GLDM / NGTDM before it waits once (prad_result_alloc +
    enqueue-only feature calls); `enqueueSegment: False` evaluates class after class as the reference does
    (featureextractor.py:560-604).  Same values, bit for bit."""
    import torch
    from pyradiomics_amd import engine
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    from pyradiomics_amd.image import Image
    rng = np.random.default_rng(5)
    N = 48
    vol = (rng.normal(size=(N, N, N)).cumsum(0).cumsum(1) * 9).astype(np.int16)
    zz, yy, xx = np.ogrid[:N, :N, :N]
    mask = (((zz - 24) ** 2 + (yy - 22) ** 2 + (xx - 25) ** 2) < 19 ** 2).astype(np.int16)
    res = {}
    for on in (True, False):
        ex = RadiomicsFeatureExtractor({"setting": {"binCount": 24, "additionalInfo": False, "enqueueSegment": on},
                                        "imageType": {"Original": {}, "Wavelet": {}}})
        res[on] = ex.execute(Image(vol), Image(mask))
    assert list(res[True].keys()) == list(res[False].keys()) and len(res[True]) > 700
    for k in res[True]:
        a, b = np.asarray(res[True][k], dtype=float), np.asarray(res[False][k], dtype=float)
        assert np.array_equal(a, b, equal_nan=True), k
    # the enqueue-only calls themselves: values arrive in the arena with the stream
    dev = torch.device("cuda", 0)
    lev = torch.from_numpy(rng.integers(1, 17, size=(20, 24, 70)).astype(np.int32)).to(dev)
    msk = torch.ones_like(lev, dtype=torch.uint8)
    g, r, _ = engine.glcm_glrlm(lev, msk, 16)
    want = engine.glcm_features(g), engine.zone_matrix_features(r, np.arange(1, r.shape[1] + 1)), engine.glcm_mcc(g)
    got = (engine.glcm_features(g, deferred=True), engine.zone_matrix_features(r, np.arange(1, r.shape[1] + 1), deferred=True),
           engine.glcm_mcc(g, deferred=True))
    engine.deferred_status()
    assert np.array_equal(got[0][0], want[0][0], equal_nan=True) and np.array_equal(got[0][1] != 0, want[0][1])
    assert np.array_equal(got[1][0], want[1][0], equal_nan=True) and np.array_equal(got[1][1] != 0, want[1][1])
    assert got[2][-1] == 0 and np.array_equal(got[2][:-1], want[2], equal_nan=True)
    # a level outside [1, Ng] under the mask voids a queued GLDM / NGTDM call: reported by the status, not lost
    bad = lev.clone()
    bad[3, 4, 5] = 40
    engine.gldm(bad, msk, 16, deferred=True)
    with pytest.raises(RuntimeError):
        engine.deferred_status()
    engine.deferred_status()                      # (the flag was cleared)
    with pytest.raises(IndexError):
        engine.gldm(bad, msk, 16)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,ng,kind", [((40, 48, 64), 16, "smooth"), ((33, 31, 70), 8, "noise"), ((1, 96, 128), 12, "smooth"),
                                           ((24, 40, 44), 5, "blocks")])
def test_glszm_features_one_queue_equals_three_calls(shape, ng, kind):
    """prad_glszm_features_dev (zones -> sizes ranked on the device -> compact matrix -> formulas, no host round trip) against
    prad_calculate_glszm_dev + prad_glszm_sizes + prad_fill_glszm_compact_dev + prad_zone_matrix_features_dev, synchronous and
    enqueue-only; zones of 8192+ voxels (the sorted 'large' list) included"""
    import torch
    from pyradiomics_amd import engine
    rng = np.random.default_rng(11)
    if kind == "noise":
        lev = rng.integers(1, ng + 1, size=shape)
    elif kind == "blocks":      # a few huge zones (> 8192 voxels) next to small ones
        lev = np.ones(shape, dtype=np.int64)
        lev[:, :20, :] = 2
        lev[:12, 20:, :22] = 3
        lev[5:9, 3:9, 3:30] = 4
        lev[rng.random(shape) < 0.01] = 5
    else:
        f = rng.normal(size=shape)
        for ax in range(3):
            if shape[ax] > 1:
                f = np.cumsum(f, axis=ax)
        lev = np.clip(((f - f.min()) / (np.ptp(f) + 1e-9) * ng).astype(np.int64) + 1, 1, ng)
    msk = rng.random(shape) < 0.93
    dev = torch.device("cuda", 0)
    L = torch.from_numpy(lev.astype(np.int32)).to(dev)
    M = torch.from_numpy(msk.astype(np.uint8)).to(dev)
    Ns = int(msk.sum())
    P, sizes = engine.glszm_compact(L, M, ng, Ns)
    want, none = engine.zone_matrix_features(P, sizes)
    got, flag = engine.glszm_features(L, M, ng, Ns)
    assert got[16] == 0 and flag[0] == 0 and not none[0]
    assert np.allclose(got[:16], want[0], rtol=1e-12, atol=0, equal_nan=True)
    got2, flag2 = engine.glszm_features(L, M, ng, Ns, deferred=True)
    engine.deferred_status()
    assert np.array_equal(got2, got, equal_nan=True) and flag2[0] == 0
    # the reference's scratch rule (nzones >= 2 Ns -> IndexError, cmatrices.c:366-373) survives the device-side route
    if kind == "noise":
        with pytest.raises(IndexError):
            engine.glszm_features(L, M, ng, max(1, Ns // 64))
        v, _ = engine.glszm_features(L, M, ng, max(1, Ns // 64), deferred=True)
        engine.deferred_status()
        assert int(v[16]) & 2


@pytest.mark.gpu
@pytest.mark.parametrize("shape,alpha,f2d", [((20, 24, 64), 0, False), ((7, 33, 40), 0, True), ((12, 18, 30), 0, False),
                                             ((16, 16, 32), 1, False), ((1, 40, 48), 0, False)])
def test_gldm_and_ngtdm_from_one_pass_equal_the_separate_calls(shape, alpha, f2d):
    """prad_calculate_gldm_ngtdm_dev (one pass over the neighbourhoods, neigh4_kernel<2>) against prad_calculate_gldm_dev and
    prad_calculate_ngtdm_dev, bit for bit; rows that are no multiple of 4 voxels and alpha != 0 take the two calls inside"""
    import torch
    from pyradiomics_amd import engine
    rng = np.random.default_rng(21)
    dev = torch.device("cuda", 0)
    lev = torch.from_numpy(rng.integers(1, 13, size=shape).astype(np.int32)).to(dev)
    msk = torch.from_numpy((rng.random(shape) < 0.8).astype(np.uint8)).to(dev)
    g, n = engine.gldm_ngtdm(lev, msk, 12, alpha, (1,), f2d, 0)
    assert torch.equal(g, engine.gldm(lev, msk, 12, alpha, (1,), f2d, 0))
    assert torch.equal(n, engine.ngtdm(lev, msk, 12, (1,), f2d, 0))
    g2, n2 = engine.gldm_ngtdm(lev, msk, 12, alpha, (1,), f2d, 0, deferred=True)
    engine.deferred_status()
    assert torch.equal(g2, g) and torch.equal(n2, n)


@pytest.mark.gpu
@pytest.mark.parametrize("setting,types", [
    ({"binWidth": 25}, {"Original": {}, "LoG": {"sigma": [1.0, 2.0]}}),
    ({"binWidth": 10, "force2D": True, "force2Ddimension": 0}, {"Original": {}, "Square": {}}),
    ({"binCount": 40, "weightingNorm": "euclidean", "gldm_a": 1}, {"Original": {}, "Wavelet": {}}),
    ({"binCount": 16, "symmetricalGLCM": False, "distances": [1, 2]}, {"Original": {}}),
    ({"binWidth": 3}, {"Original": {}}),                       # ~300 grey levels: the byte-packed kernels decline
    ({"binCount": 70}, {"Original": {}, "Exponential": {}}),   # more levels than the device MCC takes
])
def test_case_pipeline_equals_class_by_class_over_settings(setting, types):
    """the queued route (enqueueSegment, default) against class after class for settings that leave the fused kernels'
    corner: binWidth binning (two synchronisations), force2D, weighting norms and gldm_a != 0 (classes that decline the
    queue), asymmetric GLCM, two distances, float32 LoG images, an ROI above 2^20 voxels (first-order queue)"""
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    from pyradiomics_amd.image import Image
    rng = np.random.default_rng(9)
    N = 112
    f = rng.normal(size=(N, N, N))
    for ax in range(3):
        f = np.cumsum(f, axis=ax)
    vol = ((f - f.min()) / np.ptp(f) * 900 + rng.normal(size=f.shape) * 6).astype(np.int16)
    zz, yy, xx = np.ogrid[:N, :N, :N]
    mask = (((zz - 56) ** 2 + (yy - 54) ** 2 + (xx - 57) ** 2) < 52 ** 2).astype(np.int16)   # ~589 k voxels
    if "Wavelet" in types:
        mask[:] = 1                                                                            # 1.4 M voxels: first-order queue
    res = {}
    for on in (True, False):
        s = dict(setting, additionalInfo=False, enqueueSegment=on)
        ex = RadiomicsFeatureExtractor({"setting": s, "imageType": types})
        res[on] = ex.execute(Image(vol, spacing=(1.0, 1.0, 1.0)), Image(mask, spacing=(1.0, 1.0, 1.0)))
    assert list(res[True].keys()) == list(res[False].keys()) and len(res[True]) >= 90
    for k in res[True]:
        a, b = np.asarray(res[True][k], dtype=float), np.asarray(res[False][k], dtype=float)
        assert np.array_equal(a, b, equal_nan=True), (k, a, b)


@pytest.mark.gpu
def test_image_enqueue_one_call_equals_the_per_class_calls_and_bounds_its_tickets():
    """prad_image_enqueue_dev (every class of a derived image queued by one call) against the per-class engine calls; at most
    four images may be in flight per thread"""
    import torch
    from pyradiomics_amd import engine
    rng = np.random.default_rng(3)
    shape = (36, 40, 64)
    dev = torch.device("cuda", 0)
    f = rng.normal(size=shape).cumsum(0).cumsum(2)
    raw = torch.from_numpy(f).to(dev)
    lev = torch.from_numpy(np.clip(((f - f.min()) / np.ptp(f) * 12).astype(np.int32) + 1, 1, 12)).to(dev)
    msk = torch.from_numpy((rng.random(shape) < 0.9).astype(np.uint8)).to(dev)
    Ns = int(msk.sum().item())
    allc = (engine.IMG_GLCM | engine.IMG_MCC | engine.IMG_GLRLM | engine.IMG_GLDM | engine.IMG_NGTDM | engine.IMG_GLSZM)
    tok = engine.image_enqueue(lev, msk, raw, 12, Ns, allc)
    assert engine.image_wait(tok)
    res, lay = tok["res"], tok["layout"]
    Na = lay[11]
    g, r, _ = engine.glcm_glrlm(lev, msk, 12)
    want, empty = engine.glcm_features(g)
    assert Na == g.shape[2] and np.array_equal(res[lay[0]:lay[0] + Na * 23].reshape(Na, 23), want, equal_nan=True)
    assert np.array_equal(res[lay[1]:lay[1] + Na].view(np.int32)[:Na] != 0, empty)
    assert res[lay[2] + Na] == 0 and np.array_equal(res[lay[2]:lay[2] + Na], engine.glcm_mcc(g), equal_nan=True)
    want, _ = engine.zone_matrix_features(r, np.arange(1, r.shape[1] + 1))
    assert np.array_equal(res[lay[3]:lay[3] + Na * 16].reshape(Na, 16), want, equal_nan=True)
    P = engine.gldm(lev, msk, 12)
    assert np.array_equal(res[lay[5]:lay[5] + 16], engine.zone_matrix_features(P, np.arange(1, P.shape[1] + 1))[0][0], equal_nan=True)
    assert np.array_equal(res[lay[7]:lay[7] + 5], engine.ngtdm_features(engine.ngtdm(lev, msk, 12)), equal_nan=True)
    assert np.array_equal(res[lay[8]:lay[8] + 17], engine.glszm_features(lev, msk, 12, Ns)[0], equal_nan=True)
    assert lay[10] == -1                       # first order was not asked for
    toks = [engine.image_enqueue(lev, msk, raw, 12, Ns, engine.IMG_NGTDM) for _ in range(4)]
    with pytest.raises(ValueError):
        engine.image_enqueue(lev, msk, raw, 12, Ns, engine.IMG_NGTDM)
    assert all(engine.image_wait(t) for t in toks)
    assert engine.image_wait(engine.image_enqueue(lev, msk, raw, 12, Ns, engine.IMG_NGTDM))


@pytest.mark.gpu
def test_case_pipeline_on_a_2d_image_and_a_small_roi():
    """one-call enqueue on a 2-D image (4 / 8 neighbours) and on a ROI far below the first-order queue's threshold"""
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    from pyradiomics_amd.image import Image
    rng = np.random.default_rng(12)
    img2 = (rng.normal(size=(96, 128)).cumsum(0).cumsum(1) * 7).astype(np.int16)
    m2 = np.zeros((96, 128), dtype=np.int16)
    m2[10:80, 20:110] = 1
    img3 = (rng.normal(size=(20, 24, 28)).cumsum(2) * 30).astype(np.int16)
    m3 = np.zeros((20, 24, 28), dtype=np.int16)
    m3[4:9, 5:11, 6:13] = 1
    for img, msk in ((img2, m2), (img3, m3)):
        res = {}
        for on in (True, False):
            ex = RadiomicsFeatureExtractor({"setting": {"binWidth": 20, "additionalInfo": False, "enqueueSegment": on},
                                            "imageType": {"Original": {}, "Square": {}}})
            res[on] = ex.execute(Image(img), Image(msk))
        assert list(res[True].keys()) == list(res[False].keys()) and len(res[True]) >= 150
        for k in res[True]:
            assert np.array_equal(np.asarray(res[True][k], dtype=float), np.asarray(res[False][k], dtype=float), equal_nan=True), k


@pytest.mark.gpu
def test_case_pipeline_is_deterministic_under_repetition_and_threads():
    """the same cases through the pipelined route again and again, on one thread and on three (batch.run_batch): every run
    gives the same values bit for bit (look-ahead, side streams and workspace sets leave no room for a race)"""
    from pyradiomics_amd import batch
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    from pyradiomics_amd.image import Image
    rng = np.random.default_rng(31)
    N = 72
    zz, yy, xx = np.ogrid[:N, :N, :N]
    mask = (((zz - 36) ** 2 + (yy - 35) ** 2 + (xx - 37) ** 2) < 33 ** 2).astype(np.int16)
    vols = [(rng.normal(size=(N, N, N)).cumsum(i % 3).cumsum((i + 1) % 3) * (5 + i)).astype(np.int16) for i in range(6)]
    ex = RadiomicsFeatureExtractor({"setting": {"binCount": 24, "additionalInfo": False},
                                    "imageType": {"Original": {}, "Wavelet": {}, "LoG": {"sigma": [1.0, 3.0]}}})

    def one(v):
        return ex.execute(Image(v), Image(mask))

    ref = [one(v) for v in vols]
    for _ in range(3):
        again = [one(v) for v in vols]
        par = batch.run_batch(vols, one, threads=3)
        for a, b, c in zip(ref, again, par):
            assert list(a.keys()) == list(b.keys()) == list(c.keys())
            for k in a:
                x = np.asarray(a[k], dtype=float)
                assert np.array_equal(x, np.asarray(b[k], dtype=float), equal_nan=True), k
                assert np.array_equal(x, np.asarray(c[k], dtype=float), equal_nan=True), k


# ---- round 4: sliding-window voxel maps (kernels_voxslide.h) against the from-scratch window kernel --------------------
def _slide_vs_window(img, msk, Ng, vox, feats, **kw):
    import os
    import torch
    from pyradiomics_amd import engine
    dev = torch.device("cuda", 0)
    args = (torch.from_numpy(img).to(dev), torch.from_numpy(msk).to(dev), Ng, torch.from_numpy(vox).to(dev), feats)
    new = {k: v.cpu().numpy() for k, v in engine.voxel_glcm_features(*args, **kw).items()}
    variant = engine.last_variant()
    os.environ["PRAD_VOX_NO_SLIDE"] = "1"
    try:
        old = {k: v.cpu().numpy() for k, v in engine.voxel_glcm_features(*args, **kw).items()}
        assert engine.last_variant() == "window"
    finally:
        del os.environ["PRAD_VOX_NO_SLIDE"]
    return new, old, variant


@pytest.mark.parametrize("shape", [(9, 37, 70), (5, 16, 130), (1, 40, 66), (23, 5, 9)])
@pytest.mark.parametrize("radius,force2D", [(2, False), (1, False), (2, True), (1, True)])
@pytest.mark.parametrize("Ng", [32, 7, 33, 45, 64])
def test_sliding_window_maps_equal_the_window_kernel(shape, radius, force2D, Ng):
    """every centre of the volume, partial mask (holes, empty border rows): JointEntropy / JointEnergy / JointAverage from
    the incrementally updated tables == the from-scratch kernel, NaN patterns included (centres without any pair, the
    plain-mean NaN rule of JointAverage)"""
    rng = np.random.default_rng(hash((shape, radius, force2D, Ng)) % (2 ** 32))
    img = rng.integers(1, Ng + 1, size=shape).astype(np.int32)
    img[:, : shape[1] // 2] = (img[:, : shape[1] // 2] + 3) // 4 + 1          # a smoother half: repeated pairs, larger counts
    msk = rng.random(shape) < 0.85
    msk[:, -3:, :] = False                                                    # centres whose window is nearly empty
    hole = msk[:, :, shape[2] // 2: shape[2] // 2 + 7]
    hole &= rng.random(hole.shape) < 0.1
    vox = np.array(np.nonzero(np.ones(shape, bool))).astype(np.int32)        # every voxel a centre, ROI or not
    feats = ["JointEntropy", "JointEnergy", "JointAverage"]
    new, old, variant = _slide_vs_window(img, msk, Ng, vox, feats, kernelRadius=radius, force2D=force2D, force2Ddimension=0)
    assert variant == "slide"
    for f in feats:
        a, b = new[f], old[f]
        assert np.array_equal(np.isnan(a), np.isnan(b)), f
        ok = ~np.isnan(a)
        assert ok.sum() > 0
        np.testing.assert_allclose(a[ok], b[ok], rtol=1e-12 if f != "JointEntropy" else 1e-11, atol=1e-13, err_msg=f)


@pytest.mark.parametrize("feats", [["JointEntropy"], ["JointEnergy"], ["JointAverage"], ["JointEntropy", "JointEnergy"]])
@pytest.mark.parametrize("radius,force2D", [(2, False), (1, False), (2, True), (1, True)])
@pytest.mark.parametrize("Ng", [32, 38, 64])
def test_sliding_window_instantiations_by_request(feats, radius, force2D, Ng):
    """round 6: the kernel is instantiated by what the request needs -- JointEntropy alone (LIGHT: 8-byte LUT entries, pairs
    counted by the lanes), without / with the sum of i + j (JA) -- and 3-D windows run the lane-balanced schedule (helper
    lanes, LDS records); a volume wider than two runs of 64 and of 32 centres, rows that are no whole group, every centre
    against the from-scratch window kernel"""
    shape = (7, 22, 150)
    rng = np.random.default_rng(hash((tuple(feats), radius, force2D, Ng, 6)) % (2 ** 32))
    img = rng.integers(1, Ng + 1, size=shape).astype(np.int32)
    img[:, : shape[1] // 2] = (img[:, : shape[1] // 2] + 3) // 4 + 1          # repeated pairs: counts beyond 1
    img[2:5, 4:9, 20:90] = Ng                                                 # a constant block: the largest counts a window has
    msk = rng.random(shape) < 0.9
    msk[:, -2:, :] = False
    msk[3, 5:8, 30:60] = True
    vox = np.array(np.nonzero(np.ones(shape, bool))).astype(np.int32)
    new, old, variant = _slide_vs_window(img, msk, Ng, vox, feats, kernelRadius=radius, force2D=force2D, force2Ddimension=0)
    assert variant == "slide"
    for f in feats:
        a, b = new[f], old[f]
        assert np.array_equal(np.isnan(a), np.isnan(b)), f
        ok = ~np.isnan(a)
        assert ok.sum() > 0
        np.testing.assert_allclose(a[ok], b[ok], rtol=1e-12 if f != "JointEntropy" else 1e-11, atol=1e-13, err_msg=f)


@pytest.mark.parametrize("shape", [(9, 37, 70), (1, 40, 66), (23, 5, 9)])
@pytest.mark.parametrize("radius,force2D", [(2, False), (1, False), (2, True)])
@pytest.mark.parametrize("Ng", [32, 40, 64])
def test_sliding_window_wide_features_equal_the_window_kernel(shape, radius, force2D, Ng):
    """round 5: the fourteen features of the WIDE instantiation (integer pair sums, exact int64 central moments, fixed-point
    g(|i - j|) sums) against the from-scratch kernel on every centre of a partially masked volume"""
    rng = np.random.default_rng(hash((shape, radius, force2D, Ng, 5)) % (2 ** 32))
    img = rng.integers(1, Ng + 1, size=shape).astype(np.int32)
    img[:, : shape[1] // 2] = (img[:, : shape[1] // 2] + 3) // 4 + 1
    msk = rng.random(shape) < 0.85
    msk[:, -3:, :] = False
    vox = np.array(np.nonzero(np.ones(shape, bool))).astype(np.int32)
    feats = ["JointEntropy", "Autocorrelation", "ClusterProminence", "ClusterShade", "ClusterTendency", "Contrast", "DifferenceAverage",
             "DifferenceVariance", "Id", "Idm", "Idn", "Idmn", "InverseVariance", "SumAverage", "SumSquares"]
    new, old, variant = _slide_vs_window(img, msk, Ng, vox, feats, kernelRadius=radius, force2D=force2D, force2Ddimension=0)
    assert variant == "slide"
    for f in feats:
        a, b = new[f], old[f]
        assert np.array_equal(np.isnan(a), np.isnan(b)), f
        ok = ~np.isnan(a)
        assert ok.sum() > 0
        np.testing.assert_allclose(a[ok], b[ok], rtol=1e-9, atol=1e-10, err_msg=f)


def test_sliding_window_maps_are_only_taken_where_they_apply():
    rng = np.random.default_rng(3)
    shape = (8, 20, 40)
    img = rng.integers(1, 9, size=shape).astype(np.int32)
    msk = np.ones(shape, bool)
    vox = np.array(np.nonzero(msk)).astype(np.int32)
    for feats, kw, want in ((["JointEntropy"], dict(kernelRadius=2), "slide"),
                            (["JointEntropy", "Contrast"], dict(kernelRadius=2), "slide"),         # (carried since round 5: WIDE)
                            (["JointEntropy", "Correlation"], dict(kernelRadius=2), "window"),     # a feature it does not carry
                            (["JointEntropy"], dict(kernelRadius=3), "window"),                    # counts beyond a byte
                            (["JointEntropy"], dict(kernelRadius=2, symmetrical=False), "window"),
                            (["JointEntropy"], dict(kernelRadius=2, force2D=True, force2Ddimension=1), "window")):
        new, old, variant = _slide_vs_window(img, msk, 8, vox, feats, **kw)
        assert variant == want, (feats, kw, variant)
        for f in feats:
            np.testing.assert_allclose(new[f], old[f], rtol=1e-11, atol=1e-13, equal_nan=True)
    few = vox[:, ::50]                                                               # sparse centres: not worth a whole map
    _, _, variant = _slide_vs_window(img, msk, 8, np.ascontiguousarray(few), ["JointEntropy"], kernelRadius=2)
    assert variant == "window"


@pytest.mark.gpu
def test_launcher_and_two_half_binning_equal_the_one_call_forms(monkeypatch):
    """round 6: prad_image_submit / _result / _wait (the image's launches issued by the calling thread's launcher thread) against
    prad_image_enqueue_dev issued by the caller itself -- same result block, value for value; a voided image (a masked level outside
    [1, Ng]) comes back False from the wait and frees its slot; a fifth job in flight is refused at once.  prad_bincount_enqueue_dev
    + prad_bincount_wait against prad_bincount_dev: levels, number of levels, edges and the level census bit for bit."""
    import torch
    from pyradiomics_amd import engine
    rng = np.random.default_rng(8)
    shape = (30, 44, 52)
    dev = torch.device("cuda", 0)
    f = rng.normal(size=shape).cumsum(1).cumsum(2)
    raw = torch.from_numpy(f).to(dev)
    msk = torch.from_numpy((rng.random(shape) < 0.85).astype(np.uint8)).to(dev)
    # the two halves of the binning
    lv1, top1, edges1, counts1 = engine.bin_image(raw, msk, with_counts=True, binCount=20)
    tok = engine.bin_image_enqueue(raw, msk, binCount=20)
    assert tok is not None
    lv2, top2, edges2, counts2 = engine.bin_image_collect(tok, with_counts=True)
    assert top1 == top2 and torch.equal(lv1, lv2) and np.array_equal(edges1, edges2) and np.array_equal(counts1, counts2)
    assert engine.bin_image_enqueue(raw, msk, binWidth=0.5) is None                     # (binWidth: the synchronous route)
    toks = [engine.bin_image_enqueue(raw, msk, binCount=8 + i) for i in range(4)]
    assert all(t is not None for t in toks) and engine.bin_image_enqueue(raw, msk, binCount=5) is None    # four tickets per thread
    for i, t in enumerate(toks):
        assert engine.bin_image_collect(t)[1] == 8 + i
    # the launcher against the caller's own launches
    Ns = int(msk.sum().item())
    allc = (engine.IMG_GLCM | engine.IMG_MCC | engine.IMG_GLRLM | engine.IMG_GLDM | engine.IMG_NGTDM | engine.IMG_GLSZM | engine.IMG_FIRSTORDER)
    assert engine._IMAGE_LAUNCHER
    a = engine.image_enqueue(lv1, msk, raw, top1, Ns, allc)
    assert a.get("job") is not None and a["res"] is None
    assert engine.image_wait(a) and engine.image_wait(a)                               # (a second wait returns the kept verdict)
    monkeypatch.setattr(engine, "_IMAGE_LAUNCHER", False)
    b = engine.image_enqueue(lv1, msk, raw, top1, Ns, allc)
    assert b.get("job") is None and engine.image_wait(b)
    monkeypatch.setattr(engine, "_IMAGE_LAUNCHER", True)
    assert a["layout"] == b["layout"]
    n = a["layout"][12]
    assert np.array_equal(a["res"][:n], b["res"][:n], equal_nan=True)
    # a voided image, then the slot is free again; five in flight are refused
    bad = lv1.clone()
    bad[tuple(torch.nonzero(msk)[0].tolist())] = top1 + 3
    v = engine.image_enqueue(bad, msk, raw, top1, Ns, engine.IMG_GLCM | engine.IMG_GLRLM)
    assert engine.image_wait(v) is False
    jobs = [engine.image_enqueue(lv1, msk, raw, top1, Ns, engine.IMG_NGTDM) for _ in range(4)]
    with pytest.raises(ValueError):
        engine.image_enqueue(lv1, msk, raw, top1, Ns, engine.IMG_NGTDM)
    assert all(engine.image_wait(t) for t in jobs)
    assert engine.image_wait(engine.image_enqueue(lv1, msk, raw, top1, Ns, engine.IMG_NGTDM))
