"""BASELINE config 1 end to end: data/brain1_image.nrrd + brain1_label.nrrd (copied into tests/golden/data as test
vectors), Original image type, binWidth 25, through RadiomicsFeatureExtractor -- on the CPU oracle backend here and
on the HIP backend in the gpu tier -- against the reference's golden feature values (1e-6 relative)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, load_baseline_features

IMG = os.path.join(GOLDEN, "data", "brain1_image.nrrd")
LBL = os.path.join(GOLDEN, "data", "brain1_label.nrrd")


def _check_against_baseline(result, classes, config="brain1"):
    want = load_baseline_features()[config]["features"]
    for cls in classes:
        for name, ref in want[cls].items():
            val = float(result["original_%s_%s" % (cls, name)])
            assert abs(val - ref) <= 1e-6 * abs(ref), (cls, name, val, ref)
    assert result["diagnostics_Mask-original_VoxelNum"] == 4137
    assert result["diagnostics_Mask-original_BoundingBox"] == (162, 84, 11, 47, 70, 7)   # baseline_glcm.csv row 3


def test_extractor_config1_on_oracle_backend(oracle_port):
    from pyradiomics_amd import backend
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    old = backend._cmatrices
    backend.set(oracle_port)
    try:
        ex = RadiomicsFeatureExtractor(binWidth=25)
        ex.disableAllFeatures()
        ex.enableFeatureClassByName("glcm")
        res = ex.execute(IMG, LBL)
        assert [k for k in res if k.startswith("original_")][0].startswith("original_glcm_")
        assert sum(k.startswith("original_glcm_") for k in res) == 24
        _check_against_baseline(res, ["glcm"])
    finally:
        backend.set(old)


def test_extractor_normalization_on_oracle_backend(oracle_port):
    """whole-image normalisation (sitk.Normalize semantics) through the extractor on the full brain1 NRRD pair against
    the reference's `brain1_normalization` golden vectors (all six classes)"""
    from pyradiomics_amd import backend
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    old = backend._cmatrices
    backend.set(oracle_port)
    try:
        res = RadiomicsFeatureExtractor(binWidth=5, normalize=True, normalizeScale=100).execute(IMG, LBL)
        _check_against_baseline(res, ["firstorder", "glcm", "glrlm", "glszm", "gldm", "ngtdm"], "brain1_normalization")
    finally:
        backend.set(old)


def test_extractor_resampling_on_oracle_backend(oracle_port):
    """resampledPixelSpacing [2, 2, 2] with the B-spline interpolator through the extractor on whole NRRD pairs against
    the reference's `_resampling` golden vectors: pins the restated ITK resampling arithmetic end to end"""
    from pyradiomics_amd import backend
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    old = backend._cmatrices
    backend.set(oracle_port)
    try:
        for case in ("brain1", "breast1"):
            img, lbl = os.path.join(GOLDEN, "data", case + "_image.nrrd"), os.path.join(GOLDEN, "data", case + "_label.nrrd")
            res = RadiomicsFeatureExtractor(resampledPixelSpacing=[2, 2, 2], interpolator="sitkBSpline", padDistance=5).execute(img, lbl)
            want = load_baseline_features()[case + "_resampling"]["features"]
            for cls in want:
                for name, ref in want[cls].items():
                    val = float(res["original_%s_%s" % (cls, name)])
                    assert abs(val - ref) <= 1e-6 * abs(ref), (case, cls, name, val, ref)
    finally:
        backend.set(old)


def test_resample_nearest_linear_and_in_plane_only():
    from pyradiomics_amd import imageoperations
    from pyradiomics_amd.image import Image
    rng = np.random.default_rng(4)
    arr = rng.integers(0, 1000, (6, 20, 24)).astype(np.int16)
    msk = np.zeros(arr.shape, dtype=np.int16)
    msk[2:5, 5:15, 6:20] = 1
    image, mask = Image(arr, (1.0, 1.0, 3.0)), Image(msk, (1.0, 1.0, 3.0))
    for interp in ("sitkNearestNeighbor", "sitkLinear", "sitkBSpline"):
        ri, rm = imageoperations.resampleImage(image, mask, resampledPixelSpacing=[2, 2, 0], interpolator=interp, padDistance=2)
        assert ri.spacing == (2.0, 2.0, 3.0) and ri.array.shape == rm.array.shape and ri.array.dtype == np.int16
        assert ri.array.shape[0] == arr.shape[0] or ri.array.shape[0] <= arr.shape[0]      # z is left alone
        assert rm.array.max() == 1 and ri.array.min() >= -500 and ri.array.max() <= 1500
    same, samem = imageoperations.resampleImage(image, mask, resampledPixelSpacing=[1, 1, 3])
    assert same.array.shape == (3, 10, 14) and np.array_equal(same.array, arr[2:5, 5:15, 6:20])   # equal spacing: a crop
    with pytest.raises(AssertionError):
        imageoperations.resampleImage(image, mask, resampledPixelSpacing=[2, 2])


def test_extractor_params_dict_and_names():
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    ex = RadiomicsFeatureExtractor({"setting": {"binWidth": 25, "force2D": True},
                                    "imageType": {"Original": {}, "LoG": {"sigma": [2.0]}, "Wavelet": {}},
                                    "featureClass": {"glcm": ["JointEntropy"], "shape": None}})
    assert ex.settings["force2D"] is True and ex.settings["padDistance"] == 5
    assert list(ex.enabledImagetypes) == ["Original", "LoG", "Wavelet"] and ex.enabledFeatures == {"glcm": ["JointEntropy"]}
    with pytest.raises(NotImplementedError):
        RadiomicsFeatureExtractor({"imageType": {"LBP3D": {}}})


@pytest.mark.gpu
def test_extractor_config1_on_gpu():
    from pyradiomics_amd import backend, cmatrices
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    backend.set(cmatrices)
    res = RadiomicsFeatureExtractor(binWidth=25).execute(IMG, LBL)      # all five texture classes
    _check_against_baseline(res, ["glcm", "glrlm", "glszm", "gldm", "ngtdm"])


@pytest.mark.gpu
def test_extractor_normalization_on_gpu():
    from pyradiomics_amd import backend, cmatrices
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    backend.set(cmatrices)
    for route in (True, False):
        res = RadiomicsFeatureExtractor(binWidth=5, normalize=True, normalizeScale=100, deviceResident=route).execute(IMG, LBL)
        _check_against_baseline(res, ["firstorder", "glcm", "glrlm", "glszm", "gldm", "ngtdm"], "brain1_normalization")
    clipped = RadiomicsFeatureExtractor(binWidth=5, normalize=True, normalizeScale=100, removeOutliers=1.5).execute(IMG, LBL)
    assert float(clipped["original_firstorder_Maximum"]) <= 150.0 and float(clipped["original_firstorder_Minimum"]) >= -150.0


@pytest.mark.gpu
def test_extractor_resampling_on_gpu():
    """the `_resampling` golden vectors through the device-resident extractor (prad_resample_dev + everything after)"""
    from pyradiomics_amd import backend, cmatrices
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    backend.set(cmatrices)
    for case in ("brain1", "breast1"):
        img, lbl = os.path.join(GOLDEN, "data", case + "_image.nrrd"), os.path.join(GOLDEN, "data", case + "_label.nrrd")
        res = RadiomicsFeatureExtractor(resampledPixelSpacing=[2, 2, 2], interpolator="sitkBSpline", padDistance=5).execute(img, lbl)
        want = load_baseline_features()[case + "_resampling"]["features"]
        for cls in want:
            for name, ref in want[cls].items():
                val = float(res["original_%s_%s" % (cls, name)])
                assert abs(val - ref) <= 1e-6 * abs(ref), (case, cls, name, val, ref)


@pytest.mark.gpu
def test_extractor_filters_feed_matrices_on_gpu():
    """config 3 in miniature: wavelet (8 sub-bands) + LoG images re-discretised and pushed through GLCM/GLRLM"""
    from pyradiomics_amd import backend, cmatrices
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    backend.set(cmatrices)
    ex = RadiomicsFeatureExtractor({"setting": {"binCount": 32},
                                    "imageType": {"Wavelet": {}, "LoG": {"sigma": [1.0, 3.0]}},
                                    "featureClass": {"glcm": ["JointEntropy", "Contrast"], "glrlm": ["RunEntropy"]}})
    res = ex.execute(IMG, LBL)
    keys = [k for k in res if not k.startswith("diagnostics")]
    assert len(keys) == (8 + 2) * 3
    assert "wavelet-LLH_glcm_JointEntropy" in res and "log-sigma-3-0-mm-3D_glrlm_RunEntropy" in res
    assert all(np.isfinite(float(res[k])) for k in keys)


@pytest.mark.gpu
def test_device_resident_route_equals_host_array_route():
    """The default route (case kept in HBM: filters -> crop -> binning -> matrices on device tensors) must give
    exactly the numbers of the reference-shaped route (numpy binning, host arrays through the operator module)."""
    from pyradiomics_amd import backend, cmatrices
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    backend.set(cmatrices)
    params = {"setting": {"binWidth": 25}, "imageType": {"Original": {}, "Wavelet": {}, "LoG": {"sigma": [2.0]}}}
    dev = RadiomicsFeatureExtractor(params).execute(IMG, LBL)
    params["setting"]["deviceResident"] = False
    host = RadiomicsFeatureExtractor(params).execute(IMG, LBL)
    keys = [k for k in host if not k.startswith("diagnostics")]
    assert len(keys) == (1 + 8 + 1) * (18 + 24 + 16 + 16 + 14 + 5) and set(keys) <= set(dev)
    for k in keys:
        a, b = float(dev[k]), float(host[k])
        # identical inputs reach both routes (levels, matrices are bit-identical); the device route also evaluates
        # the feature formulas on the GPU, which reorders float sums
        assert a == b or (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-10 * abs(b), (k, a, b)
    assert dev["diagnostics_Mask-original_BoundingBox"] == host["diagnostics_Mask-original_BoundingBox"]
    params["setting"]["deviceResident"] = True
    params["setting"]["fusedSegment"] = False          # device-resident matrices + numpy formulas: bit-identical
    mid = RadiomicsFeatureExtractor(params).execute(IMG, LBL)
    for k in keys:
        a, b = float(mid[k]), float(host[k])
        if "_firstorder_" in k:
            # the device crop pads its rows to a multiple of 4 voxels (outside the ROI): the first-order block
            # reductions partition a differently shaped array, i.e. the same terms are summed in another order
            assert a == b or (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-13 * abs(b), (k, a, b)
        else:
            assert a == b or (np.isnan(a) and np.isnan(b)), (k, a, b)


@pytest.mark.gpu
def test_device_resident_binning_modes_and_resegmentation():
    from pyradiomics_amd import backend, cmatrices
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    backend.set(cmatrices)
    for setting in ({"binCount": 16}, {"binWidth": 10, "force2D": True},
                    {"binWidth": 25, "resegmentRange": [-1, 1], "resegmentMode": "sigma"}, {"binWidth": 25, "distances": [1, 2]}):
        dev = RadiomicsFeatureExtractor({"setting": dict(setting)}).execute(IMG, LBL)
        host = RadiomicsFeatureExtractor({"setting": dict(setting, deviceResident=False)}).execute(IMG, LBL)
        for k in host:
            if not k.startswith("diagnostics"):
                a, b = float(dev[k]), float(host[k])
                assert a == b or (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-10 * abs(b), (setting, k, a, b)


@pytest.mark.gpu
def test_voxel_based_extraction_all_classes_vs_oracle_backend(oracle_port):
    """whole voxel-based extraction (exampleVoxel.yaml-style parameters, all six classes): fused on-device feature
    maps of the GPU backend against the reference's route (per-kernel matrices + numpy formulas) on the CPU oracle"""
    from pyradiomics_amd import backend, cmatrices
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    from helpers import load_case
    image, mask, _ = load_case("breast1")
    params = {"setting": {"binWidth": 25, "force2D": True, "label": 1},
              "voxelSetting": {"kernelRadius": 2, "maskedKernel": True, "initValue": float("nan"), "voxelBatch": 60},
              "featureClass": {"firstorder": ["Mean", "Entropy", "90Percentile"], "glcm": ["JointEntropy", "Idm"],
                               "glrlm": ["RunEntropy", "ShortRunEmphasis"], "glszm": ["ZonePercentage", "ZoneEntropy"],
                               "gldm": ["DependenceEntropy"], "ngtdm": ["Coarseness", "Busyness"]}}
    maps = {}
    for name, be in (("gpu", cmatrices), ("cpu", oracle_port)):
        backend.set(be)
        try:
            res = RadiomicsFeatureExtractor(params).execute(image, mask, voxelBased=True)
            maps[name] = {k: v.array for k, v in res.items() if hasattr(v, "array")}
        finally:
            backend.set(cmatrices)
    assert len(maps["gpu"]) == 12 and set(maps["gpu"]) == set(maps["cpu"])
    for k, want in maps["cpu"].items():
        got = maps["gpu"][k]
        assert got.shape == want.shape and np.array_equal(np.isnan(got), np.isnan(want)), k
        ok = ~np.isnan(want)
        assert ok.sum() == int((mask.array == 1).sum())
        np.testing.assert_allclose(got[ok], want[ok], rtol=1e-9, atol=1e-12, err_msg=k)


@pytest.mark.gpu
def test_typical_mr_parameter_set_runs_end_to_end(tmp_path):
    """the shape of the reference's examples/exampleSettings/exampleMR_*.yaml: normalise, resample to 2 mm, Original + LoG +
    Wavelet, first order + four texture classes, voxelArrayShift -- through the parameter-file constructor"""
    from pyradiomics_amd import backend, cmatrices
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    backend.set(cmatrices)
    p = tmp_path / "mr.yaml"
    p.write_text("imageType:\n  Original: {}\n  LoG:\n    sigma: [2.0, 3.0]\n  Wavelet: {}\n"
                 "featureClass:\n  shape:\n  firstorder:\n  glcm:\n    - 'JointEntropy'\n    - 'Idm'\n  glrlm:\n  glszm:\n  gldm:\n"
                 "setting:\n  normalize: true\n  normalizeScale: 100\n  interpolator: 'sitkBSpline'\n"
                 "  resampledPixelSpacing: [2, 2, 2]\n  binWidth: 5\n  voxelArrayShift: 300\n  label: 1\n")
    res = RadiomicsFeatureExtractor(str(p)).execute(IMG, LBL)
    keys = [k for k in res if not k.startswith("diagnostics")]
    assert len(keys) == (1 + 2 + 8) * (18 + 2 + 16 + 16 + 14)
    assert all(np.isfinite(float(res[k])) for k in keys), [k for k in keys if not np.isfinite(float(res[k]))][:5]
    assert res["diagnostics_Mask-original_VoxelNum"] == 4137 and res["diagnostics_Mask-interpolated_VoxelNum"] == 1915
    assert res["diagnostics_Image-interpolated_Spacing"] == (2.0, 2.0, 2.0)
    # the wavelet approximation of a normalised image keeps its energy ordering: LLL dominates the detail bands
    assert float(res["wavelet-LLL_firstorder_Energy"]) > float(res["wavelet-HHH_firstorder_Energy"])


def test_precrop_leaves_original_image_features_unchanged(oracle_port):
    """preCrop only changes what filters see: Original-image features are those of the uncropped run"""
    from pyradiomics_amd import backend
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    old = backend._cmatrices
    backend.set(oracle_port)
    try:
        kw = dict(binWidth=25)
        a = RadiomicsFeatureExtractor({"setting": dict(kw), "featureClass": {"glcm": None, "firstorder": None}}).execute(IMG, LBL)
        b = RadiomicsFeatureExtractor({"setting": dict(kw, preCrop=True, padDistance=3), "featureClass": {"glcm": None, "firstorder": None}}).execute(IMG, LBL)
        for k in a:
            if not k.startswith("diagnostics"):
                assert float(a[k]) == float(b[k]), k
    finally:
        backend.set(old)


@pytest.mark.gpu
def test_two_dimensional_image_through_the_extractor():
    """a 2-D slice (Nd = 2) as input: Original + Wavelet (4 sub-bands) + LoG, all six classes, device and host routes"""
    from pyradiomics_amd import backend, cmatrices
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    from pyradiomics_amd.image import Image, read_nrrd
    backend.set(cmatrices)
    vol, lab = read_nrrd(IMG), read_nrrd(LBL)
    sl = int(np.argmax((lab.array == 1).sum((1, 2))))
    image, mask = Image(vol.array[sl], vol.spacing[:2]), Image(lab.array[sl], lab.spacing[:2])
    params = {"setting": {"binWidth": 25}, "imageType": {"Original": {}, "Wavelet": {}, "LoG": {"sigma": [2.0]}}}
    dev = RadiomicsFeatureExtractor(params).execute(image, mask)
    keys = [k for k in dev if not k.startswith("diagnostics")]
    assert sorted({k.split("_")[0] for k in keys}) == ["log-sigma-2-0-mm-3D", "original", "wavelet-HH", "wavelet-HL", "wavelet-LH", "wavelet-LL"]
    assert len(keys) == 6 * (18 + 24 + 16 + 16 + 14 + 5) and all(np.isfinite(float(dev[k])) for k in keys)
    params["setting"]["deviceResident"] = False
    host = RadiomicsFeatureExtractor(params).execute(image, mask)
    for k in keys:
        a, b = float(dev[k]), float(host[k])
        assert a == b or abs(a - b) <= 1e-10 * abs(b), (k, a, b)
    # the same slice embedded as a 1 x Ny x Nx volume with force2D gives the same texture features
    vol3 = RadiomicsFeatureExtractor({"setting": {"binWidth": 25, "force2D": True}}).execute(
        Image(vol.array[sl][None], vol.spacing), Image(lab.array[sl][None], lab.spacing))
    for key in ("original_glcm_JointEntropy", "original_glrlm_RunEntropy", "original_glszm_ZoneEntropy",
                "original_gldm_DependenceEntropy", "original_ngtdm_Coarseness", "original_firstorder_Mean"):
        assert float(vol3[key]) == pytest.approx(float(dev[key]), rel=1e-12), key


def test_mask_geometry_check_and_correct_mask(oracle_port):
    """imageoperations.checkMask step 1 / _correctMask / _checkROI: a mask on another grid is an error unless
    correctMask resamples it (nearest neighbour) onto the image grid; geometryTolerance widens the comparison"""
    from pyradiomics_amd import backend, imageoperations as io
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    from pyradiomics_amd.image import Image
    rng = np.random.default_rng(0)
    sp, org = (0.8, 0.9, 2.0), (5.0, -3.0, 7.0)
    img = Image(rng.integers(0, 200, (10, 12, 14)).astype(np.int16), spacing=sp, origin=org)
    sub = (rng.random((6, 8, 9)) < 0.6).astype(np.uint8)
    full = np.zeros((10, 12, 14), np.uint8)
    full[2:8, 3:11, 4:13] = sub
    # (a) a cropped segmentation on the same lattice
    crop = Image(sub, spacing=sp, origin=(org[0] + 4 * sp[0], org[1] + 3 * sp[1], org[2] + 2 * sp[2]))
    with pytest.raises(ValueError, match="size mismatch"):
        io.checkMaskGeometry(img, crop)
    fixed = io.checkMaskGeometry(img, crop, correctMask=True)
    assert np.array_equal(fixed.array, full) and fixed.GetOrigin() == img.GetOrigin() and fixed.shape == img.shape
    # (b) the same crop stored with x and y swapped (a rotated direction matrix): the general resampling path
    rot = Image(np.ascontiguousarray(sub.transpose(0, 2, 1)), spacing=(sp[1], sp[0], sp[2]), origin=crop.GetOrigin(),
                direction=(0, 1, 0, 1, 0, 0, 0, 0, 1))
    assert np.array_equal(io.checkMaskGeometry(img, rot, correctMask=True).array, full)
    # (c) a finer mask grid (half the spacing in-plane): nearest neighbour picks the voxel under each image centre
    fine = np.repeat(np.repeat(full, 2, axis=1), 2, axis=2)
    finem = Image(fine, spacing=(sp[0] / 2, sp[1] / 2, sp[2]), origin=(org[0] - sp[0] / 4, org[1] - sp[1] / 4, org[2]))
    assert np.array_equal(io.checkMaskGeometry(img, finem, correctMask=True).array, full)
    # (d) tolerance: 1e-5 mm off is a mismatch at ITK's default 1e-6, fine at 1e-4
    off = Image(full, spacing=sp, origin=(org[0] + 1e-5, org[1], org[2]))
    with pytest.raises(ValueError, match="geometry mismatch"):
        io.checkMaskGeometry(img, off)
    assert io.checkMaskGeometry(img, off, geometryTolerance=1e-4) is off
    # (e) an ROI that sticks out of the image cannot be corrected; neither can a mask without the label
    out = Image(np.ones((6, 8, 9), np.uint8), spacing=sp, origin=(org[0] + 8 * sp[0], org[1], org[2]))
    with pytest.raises(ValueError, match="larger than image space"):
        io.checkMaskGeometry(img, out, correctMask=True)
    with pytest.raises(ValueError, match="not present"):
        io.checkMaskGeometry(img, Image(np.zeros((6, 8, 9), np.uint8), spacing=sp, origin=crop.GetOrigin()), correctMask=True)
    # through the extractor: the corrected crop gives the features of the full-grid mask
    backend.set(oracle_port)
    try:
        params = {"setting": {"binWidth": 25, "correctMask": True, "additionalInfo": False, "deviceResident": False},
                  "featureClass": {"glcm": ["JointEntropy", "Contrast"], "glszm": ["ZonePercentage"], "firstorder": ["Mean"]}}
        a = RadiomicsFeatureExtractor(params).execute(img, crop)
        b = RadiomicsFeatureExtractor(params).execute(img, Image(full, spacing=sp, origin=org))
        assert list(a) == list(b) and all(float(a[k]) == float(b[k]) for k in a) and len(a) == 4
        params["setting"]["correctMask"] = False
        with pytest.raises(ValueError):
            RadiomicsFeatureExtractor(params).execute(img, crop)
    finally:
        from pyradiomics_amd import cmatrices
        backend.set(cmatrices)


def test_vector_mask_and_label_channel(oracle_port, tmp_path):
    """imageoperations.getMask (:12-64): a segmentation stored as a vector image (overlapping segments, one channel
    each) -- the channel picked by label_channel is used like a scalar mask"""
    from pyradiomics_amd import backend, imageoperations as io
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    from pyradiomics_amd.image import Image, read_image
    rng = np.random.default_rng(3)
    shape = (6, 9, 11)
    img = Image(rng.integers(0, 300, shape).astype(np.int16), spacing=(0.8, 0.9, 2.0), origin=(1.0, 2.0, 3.0))
    seg = np.stack([(rng.random(shape) < 0.5), (rng.random(shape) < 0.4)], axis=-1).astype(np.uint8)   # (z, y, x, c)
    path = tmp_path / "seg.nrrd"
    header = ("NRRD0004\ntype: uchar\ndimension: 4\nspace: left-posterior-superior\nsizes: 2 11 9 6\n"
              "space directions: none (0.8,0,0) (0,0.9,0) (0,0,2)\nkinds: list domain domain domain\nencoding: raw\n"
              "space origin: (1,2,3)\n\n")
    path.write_bytes(header.encode("ascii") + seg.tobytes())
    vec = read_image(str(path))
    assert vec.components == 2 and vec.shape == shape + (2,) and vec.GetSpacing() == (0.8, 0.9, 2.0)
    assert vec.GetOrigin() == (1.0, 2.0, 3.0)
    for ch in (0, 1):
        m = io.getMask(vec, label_channel=ch)
        assert m.shape == shape and np.array_equal(m.array, seg[..., ch])
    with pytest.raises(ValueError, match="only contains 2 objects"):
        io.getMask(vec, label_channel=2)
    with pytest.raises(ValueError, match="Choose from"):
        io.getMask(vec, label=7)
    with pytest.raises(ValueError, match="nothing is segmented"):
        io.getMask(Image(np.zeros(shape, np.uint8)))
    backend.set(oracle_port)
    try:
        params = {"setting": {"binWidth": 25, "additionalInfo": False, "deviceResident": False},
                  "featureClass": {"glrlm": ["RunEntropy"], "firstorder": ["Median"]}}
        ex = RadiomicsFeatureExtractor(params)
        a = ex.execute(img, str(path), label_channel=1)
        b = ex.execute(img, Image(seg[..., 1].copy(), spacing=(0.8, 0.9, 2.0), origin=(1.0, 2.0, 3.0)))
        c = ex.execute(img.array, seg, label_channel=0)          # plain arrays, trailing component axis
        d = ex.execute(img.array, seg[..., 0].copy())
        assert all(float(a[k]) == float(b[k]) for k in b) and all(float(c[k]) == float(d[k]) for k in d) and len(a) == 2
    finally:
        from pyradiomics_amd import cmatrices
        backend.set(cmatrices)


def test_parameter_validation_follows_the_reference_schema():
    """radiomics/schemas/paramSchema.yaml + schemaFuncs.py restated in pyradiomics_amd/paramcheck.py"""
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor as Ex
    good = {"setting": {"binWidth": 25, "label": 1, "interpolator": "sitkBSpline", "resampledPixelSpacing": [2, 2, 0],
                        "weightingNorm": None, "distances": [1, 2], "resegmentRange": [-3, 3], "resegmentMode": "sigma",
                        "force2D": True, "force2Ddimension": 0, "normalize": True, "normalizeScale": 100,
                        "geometryTolerance": 1e-4, "correctMask": True, "deviceResident": False},
            "voxelSetting": {"kernelRadius": 2, "maskedKernel": True, "initValue": "nan", "voxelBatch": 10000},
            "imageType": {"Original": {}, "LoG": {"sigma": [1.0, 3]}, "Wavelet": {"wavelet": "coif1", "level": 1}},
            "featureClass": {"glcm": ["JointEntropy"], "firstorder": None, "shape": None}}
    Ex(good)
    bad = [
        {"setting": {"binWidth": 0}}, {"setting": {"binWidth": "wide"}}, {"setting": {"binCount": 2.5}},
        {"setting": {"label": 0}}, {"setting": {"normalize": 1}}, {"setting": {"minimumROIDimensions": 4}},
        {"setting": {"distances": [0]}}, {"setting": {"distances": 1}}, {"setting": {"resegmentMode": "percent"}},
        {"setting": {"interpolator": "sitkCubic"}}, {"setting": {"interpolator": 11}},
        {"setting": {"weightingNorm": "chebyshev"}}, {"setting": {"force2Ddimension": 3}},
        {"setting": {"geometryTolerance": 0}}, {"setting": {"wavelet": "db9"}},
        {"voxelSetting": {"kernelRadius": 0}}, {"voxelSetting": {"maskedKernel": "yes"}},
        {"imageType": {"LoG": {"sigma": [0.0]}}}, {"imageType": None}, {"featureClass": None},
        {"featureClass": {"glcm": "JointEntropy"}}, {"featureClass": {"glcm": ["NoSuchFeature"]}},
        {"settings": {"binWidth": 25}},
    ]
    for params in bad:
        with pytest.raises(ValueError):
            Ex(params)
    with pytest.raises(NotImplementedError):
        Ex({"imageType": {"LBP3D": {}}})                       # known to the reference, outside this package


def test_reference_example_parameter_files_are_accepted():
    """every examples/exampleSettings/*.yaml of the reference configures the extractor (shape classes are skipped)"""
    import glob
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    files = sorted(glob.glob("/root/reference/examples/exampleSettings/*.yaml"))
    if not files:
        pytest.skip("reference checkout not present")
    for f in files:
        ex = RadiomicsFeatureExtractor(f)
        assert ex.enabledImagetypes and "shape" not in ex.enabledFeatures


def test_every_feature_is_documented_with_its_reference_location():
    """mirror of the reference's tests/test_docstrings.py: every get<Name>FeatureValue carries a formula docstring that
    cites the reference method it restates"""
    import re
    from pyradiomics_amd.featureextractor import getFeatureClasses
    n = 0
    for cname, cls in getFeatureClasses().items():
        for feature in cls.getFeatureNames():
            doc = getattr(cls, "get%sFeatureValue" % feature).__doc__
            assert doc and doc.strip(), (cname, feature)
            assert re.search(r"\(%s\.py:\d+(-\d+)?\)" % cname, doc), (cname, feature, doc)
            n += 1
    assert n == 100


def test_check_mask_entry_point():
    """imageoperations.checkMask (:177-312): bounding box in the reference's (L_x, U_x, L_y, U_y, L_z, U_z) order and
    the ROI constraints"""
    from pyradiomics_amd import imageoperations as io
    from pyradiomics_amd.image import Image
    img = Image(np.zeros((6, 8, 10), np.int16))
    m = np.zeros((6, 8, 10), np.uint8)
    m[1:4, 2:7, 3:9] = 1
    bb, corrected = io.checkMask(img, Image(m))
    assert list(bb) == [3, 8, 2, 6, 1, 3] and corrected is None
    with pytest.raises(ValueError, match="not present"):
        io.checkMask(img, Image(m), label=2)
    line = np.zeros_like(m)
    line[2, 3, 1:6] = 1
    with pytest.raises(ValueError, match="too few dimensions"):
        io.checkMask(img, Image(line))
    assert list(io.checkMask(img, Image(line), minimumROIDimensions=1)[0]) == [1, 5, 3, 3, 2, 2]
    one = np.zeros_like(m)
    one[2, 3, 4] = 1
    with pytest.raises(ValueError, match="1 segmented voxel"):
        io.checkMask(img, Image(one), minimumROIDimensions=1)
    with pytest.raises(ValueError, match="too small"):
        io.checkMask(img, Image(m), minimumROISize=90)
    sub = Image(m[1:4].copy(), origin=(0.0, 0.0, 1.0))
    bb, corrected = io.checkMask(img, sub, correctMask=True)
    assert list(bb) == [3, 8, 2, 6, 1, 3] and corrected is not None and np.array_equal(corrected.array, m)


@pytest.mark.gpu
def test_exception_between_queue_and_collect_leaves_no_ticket(monkeypatch):
    """the look-ahead pipeline holds a ticket of the library's 4-slot in-flight table per queued image;
    an exception raised while the NEXT derived image is prepared (or by a host class) must retire it, or the persistent
    batch worker thread is poisoned ('4 images are in flight')"""
    from pyradiomics_amd import backend, cmatrices, imageoperations
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    backend.set(cmatrices)
    ex = RadiomicsFeatureExtractor(binWidth=25)
    ex.enableImageTypeByName("Wavelet")
    want = ex.execute(IMG, LBL)
    real = imageoperations.cropToTumorMask
    calls = {"n": 0}

    def flaky(*a, **k):
        calls["n"] += 1
        if calls["n"] == 3:                       # image 3 fails while image 2 is queued and image 1 collected
            raise RuntimeError("injected")
        return real(*a, **k)
    for _ in range(6):                             # more failures than the table has slots
        calls["n"] = 0
        monkeypatch.setattr(imageoperations, "cropToTumorMask", flaky)
        with pytest.raises(RuntimeError, match="injected"):
            ex.execute(IMG, LBL)
        monkeypatch.setattr(imageoperations, "cropToTumorMask", real)
    # a host-side class failing in _finishFeatures while its image's device classes are queued
    from pyradiomics_amd import firstorder
    real_init = firstorder.RadiomicsFirstOrder._initCalculation
    boom = {"n": 0}

    def bad(self, *a, **k):
        boom["n"] += 1
        if boom["n"] == 2:
            raise RuntimeError("injected-host")
        return real_init(self, *a, **k)
    for _ in range(6):
        boom["n"] = 0
        monkeypatch.setattr(firstorder.RadiomicsFirstOrder, "_initCalculation", bad)
        with pytest.raises(RuntimeError, match="injected-host"):
            ex.execute(IMG, LBL)
        monkeypatch.setattr(firstorder.RadiomicsFirstOrder, "_initCalculation", real_init)
    for _ in range(5):
        got = ex.execute(IMG, LBL)
        for k, v in want.items():
            if not k.startswith("diagnostics"):
                assert float(got[k]) == float(v), k
