"""shared test helpers (CPU side)"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CLASSES = ["glcm", "glrlm", "glszm", "gldm", "ngtdm"]          # the matrix-building classes
FEATURE_CLASSES = ["firstorder"] + CLASSES                      # everything with golden feature values


def feature_class(name):
    import importlib
    mod = importlib.import_module("pyradiomics_amd." + name)
    return getattr(mod, "RadiomicsFirstOrder" if name == "firstorder" else "Radiomics" + name.upper())


def load_case(case):
    from pyradiomics_amd.image import Image
    d = np.load(os.path.join(GOLDEN, case + ".npz"))
    image = Image(d["image"], spacing=d["spacing"])
    mask = Image(d["mask"].astype(np.int32), spacing=d["spacing"])
    return image, mask, {c: d["P_" + c] for c in CLASSES}


def load_baseline_features():
    with open(os.path.join(GOLDEN, "baseline_features.json")) as f:
        return json.load(f)


def prepared_case(cfg):
    """image / mask of a baseline configuration after the numpy-only preprocessing the reference's test harness
    applies (tests/testUtils.py: resegmentation then crop to the new ROI); returns (image, mask, settings)"""
    from pyradiomics_amd import imageoperations
    settings = {k: v for k, v in cfg["settings"].items() if v is not None}
    if settings.get("resampledPixelSpacing") is not None:
        # needs the whole grid (the B-spline prefilter is global): only the cases whose NRRD pair is kept in the repo
        from pyradiomics_amd.image import read_nrrd
        path = os.path.join(GOLDEN, "data", cfg["case"] + "_%s.nrrd")
        if not os.path.exists(path % "image"):
            import pytest
            pytest.skip("full image of %s is not part of the fixtures" % cfg["case"])
        image, mask = imageoperations.resampleImage(read_nrrd(path % "image"), read_nrrd(path % "label"), **settings)
        image, mask = imageoperations.cropToTumorMask(image, mask, settings.get("label", 1))
        for k in ("resampledPixelSpacing", "interpolator", "padDistance", "normalize", "normalizeScale", "removeOutliers"):
            settings.pop(k, None)
        return image, mask, settings
    image, mask, _ = load_case(cfg["case"])
    if settings.pop("normalize", False):
        # the reference normalises the WHOLE image before cropping (imageoperations.py:615-654); the fixtures hold the
        # ROI crop plus the whole-image mean / sigma, which is all normalizeImage depends on
        d = np.load(os.path.join(GOLDEN, cfg["case"] + ".npz"))
        arr = (image.array.astype(np.float64) - float(d["image_mean"])) / float(d["image_sigma"])
        if settings.get("removeOutliers") is not None:
            arr = np.clip(arr, -settings["removeOutliers"], settings["removeOutliers"])
        image = image.like(arr * float(settings.get("normalizeScale", 1)))
    for k in ("normalizeScale", "removeOutliers", "interpolator", "padDistance"):
        settings.pop(k, None)
    if cfg["settings"].get("resegmentRange") is not None:
        mask = imageoperations.resegmentMask(image, mask, **settings)
        image, mask = imageoperations.cropToTumorMask(image, mask, settings.get("label", 1))
    return image, mask, settings
