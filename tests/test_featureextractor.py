"""BASELINE config 1 end to end: data/brain1_image.nrrd + brain1_label.nrrd (copied into tests/golden/data as test
vectors), Original image type, binWidth 25, through RadiomicsFeatureExtractor -- on the CPU oracle backend here and
on the HIP backend in the gpu tier -- against the reference's golden feature values (1e-6 relative)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, load_baseline_features

IMG = os.path.join(GOLDEN, "data", "brain1_image.nrrd")
LBL = os.path.join(GOLDEN, "data", "brain1_label.nrrd")


def _check_against_baseline(result, classes):
    want = load_baseline_features()["brain1"]["features"]
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


def test_extractor_params_dict_and_names():
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    ex = RadiomicsFeatureExtractor({"setting": {"binWidth": 25, "force2D": True},
                                    "imageType": {"Original": {}, "LoG": {"sigma": [2.0]}, "Wavelet": {}},
                                    "featureClass": {"glcm": ["JointEntropy"], "shape": None}})
    assert ex.settings["force2D"] is True and ex.settings["padDistance"] == 5
    assert list(ex.enabledImagetypes) == ["Original", "LoG", "Wavelet"] and ex.enabledFeatures == {"glcm": ["JointEntropy"]}
    with pytest.raises(NotImplementedError):
        RadiomicsFeatureExtractor({"imageType": {"Gradient": {}}})


@pytest.mark.gpu
def test_extractor_config1_on_gpu():
    from pyradiomics_amd import backend, cmatrices
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    backend.set(cmatrices)
    res = RadiomicsFeatureExtractor(binWidth=25).execute(IMG, LBL)      # all five texture classes
    _check_against_baseline(res, ["glcm", "glrlm", "glszm", "gldm", "ngtdm"])


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
