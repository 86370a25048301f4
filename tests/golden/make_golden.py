#!/usr/bin/env python
"""Generates the committed golden fixtures from the reference checkout (run in the dev container only;
/root/reference does not exist on the GPU box).

For the three bundled cases whose images are present (brain1, brain2, breast1 -- lung1/lung2 images are
missing large blobs) this stores, per case, in tests/golden/<case>.npz:
    image, mask      ROI-cropped arrays (z, y, x), mask = (label == 1)            [data/<case>_image.nrrd, _label.nrrd]
    spacing          (x, y, z) voxel spacing
    P_glcm .. P_ngtdm  the reference's golden matrices data/baseline/<case>_<class>.npy  (tests/test_matrices.py:35-65)
and in tests/golden/baseline_features.json the reference's golden feature values
data/baseline/baseline_<class>.csv (tests/test_features.py) for the configurations that need no SimpleITK-only
preprocessing (default, _2d, _FBN, _combined, _flatRegion, _resegmentation, _normalization, _resampling), together with
their settings.  The whole NRRD pairs of brain1 and breast1 are copied to tests/golden/data for the tests that need the
full grid (resampling, extractor, CLI).

Nothing here is reference SOURCE; these are its test vectors."""
import ast
import csv
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from pyradiomics_amd.image import read_nrrd  # noqa: E402

REF = os.environ.get("REFERENCE", "/root/reference")
CASES = ["brain1", "brain2", "breast1"]
CLASSES = ["glcm", "glrlm", "glszm", "gldm", "ngtdm"]
CONFIG_SUFFIXES = ["", "_2d", "_FBN", "_combined", "_flatRegion", "_resegmentation", "_normalization", "_resampling"]
FULL_IMAGES = ["brain1", "breast1"]      # whole NRRD pairs kept under tests/golden/data (resampling needs the full grid)
FEATURE_CLASSES = CLASSES + ["firstorder"]      # golden feature values only (first order has no matrix)
KEEP = ("binWidth", "binCount", "force2D", "force2Ddimension", "distances", "weightingNorm", "symmetricalGLCM",
        "gldm_a", "label", "resegmentRange", "resegmentMode", "voxelArrayShift", "normalize", "normalizeScale",
        "removeOutliers", "resampledPixelSpacing", "interpolator", "padDistance")


def main():
    feats = {}
    for case in CASES:
        img = read_nrrd(os.path.join(REF, "data", case + "_image.nrrd"))
        lab = read_nrrd(os.path.join(REF, "data", case + "_label.nrrd"))
        m = lab.array == 1
        idx = np.where(m)
        sl = tuple(slice(int(i.min()), int(i.max()) + 1) for i in idx)
        # whole-image mean / standard deviation (N - 1): what sitk.Normalize uses for the `_normalization` configs
        # (imageoperations.py:615-654 normalises over ALL voxels, not just the ROI)
        x = img.array.astype(np.float64)
        mean = x.mean()
        sigma = np.sqrt(((x - mean) ** 2).sum() / (x.size - 1))
        out = {"image": img.array[sl], "mask": m[sl].astype(np.uint8), "spacing": np.array(img.spacing),
               "image_mean": np.float64(mean), "image_sigma": np.float64(sigma)}
        for cls in CLASSES:
            out["P_" + cls] = np.load(os.path.join(REF, "data", "baseline", "%s_%s.npy" % (case, cls)))
        np.savez_compressed(os.path.join(HERE, case + ".npz"), **out)
        print(case, out["image"].shape, out["image"].dtype, int(m.sum()), "voxels")
    import shutil
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    for case in FULL_IMAGES:
        for kind in ("image", "label"):
            dst = os.path.join(HERE, "data", "%s_%s.nrrd" % (case, kind))
            if not os.path.exists(dst):
                shutil.copyfile(os.path.join(REF, "data", "%s_%s.nrrd" % (case, kind)), dst)
                os.chmod(dst, 0o644)
    for cls in FEATURE_CLASSES:
        rows = list(csv.reader(open(os.path.join(REF, "data", "baseline", "baseline_%s.csv" % cls))))
        hdr = rows[0]
        byname = {r[0]: r for r in rows}
        for col in range(1, len(hdr)):
            cfg = hdr[col]
            case = cfg.split("_")[0]
            if case not in CASES or cfg[len(case):] not in CONFIG_SUFFIXES:
                continue
            settings = ast.literal_eval(byname["diagnostics_Configuration_Settings"][col])
            entry = feats.setdefault(cfg, {"case": case, "settings": {k: v for k, v in settings.items() if k in KEEP},
                                           "features": {}})
            entry["settings"].update({k: v for k, v in settings.items() if k in KEEP})
            prefix = "original_%s_" % cls
            entry["features"][cls] = {r[0][len(prefix):]: float(r[col]) for r in rows
                                      if r[0].startswith(prefix) and r[col] != ""}
    with open(os.path.join(HERE, "baseline_features.json"), "w") as f:
        json.dump(feats, f, indent=1, sort_keys=True)
    print("configs:", sorted(feats))


if __name__ == "__main__":
    main()
