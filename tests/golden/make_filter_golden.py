#!/usr/bin/env python
"""Generator of the golden vectors that PIN the filter stack to the arithmetic the
reference really executes: PyWavelets' `pywt.swtn` (radiomics/imageoperations.py:899-970) and SimpleITK's
`LaplacianRecursiveGaussianImageFilter` (:756-836).  Neither wheel is installable in the build container (no
network), so this script has not run yet; the filters are pinned meanwhile by the outputs the reference recorded in
its notebook (tests/golden/make_notebook_golden.py, tests/test_notebook_pin.py) and this file would add volume-level vectors.  Run it once in ANY environment that has numpy + PyWavelets + SimpleITK:

    python tests/golden/make_filter_golden.py            # writes tests/golden/filters_golden.npz (a few hundred KB)

and commit the file: tests/test_oracle.py::test_filters_restatement_matches_wheels (CPU tier) then pins
oracle/filters_oracle.py, and tests/test_gpu_filters.py::test_filters_match_wheel_golden pins the HIP kernels.
Inputs: the brain1 ROI crop kept in tests/golden/brain1.npz (int16, its spacing) and a seeded 32 x 30 x 28 volume
with anisotropic spacing; outputs: the 8 sub-bands of the default wavelet (coif1, level 1), a 2-level db2 set on the
seeded volume, and LoG at sigma 1, 2, 3, 5 mm."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def wavelet_subbands(array, wavelet, level, start_level=0):
    """exactly what imageoperations._swt3 does with pywt.swtn (padding to even sizes, one level at a time)"""
    import pywt
    data = array.astype(np.float64)
    original_shape = data.shape
    adjust = np.remainder(original_shape, 2).astype(int)
    pad = list(zip(np.zeros(3, dtype=int), adjust))
    data = np.pad(data, pad, "wrap")          # imageoperations.py:914-919
    out = {}
    for i in range(level):
        dec = pywt.swtn(data, wavelet, level=1, start_level=0, axes=(0, 1, 2))[0]      # keys 'aaa', 'aad', ...
        data = dec["aaa"].copy()
        for k, v in dec.items():
            name = k.replace("a", "L").replace("d", "H")
            if name == "LLL" and i < level - 1:
                continue
            v = v[tuple(slice(None, -1 if a else None) for a in adjust)]
            if i >= start_level:
                out["level%d_%s" % (i + 1, name)] = v
    return out


def log_images(array, spacing_xyz, sigmas):
    import SimpleITK as sitk
    im = sitk.GetImageFromArray(array)
    im.SetSpacing(tuple(float(s) for s in spacing_xyz))
    out = {}
    for s in sigmas:
        f = sitk.LaplacianRecursiveGaussianImageFilter()
        f.SetNormalizeAcrossScale(True)
        f.SetSigma(float(s))
        out["sigma_%g" % s] = sitk.GetArrayFromImage(f.Execute(sitk.Cast(im, sitk.sitkFloat32)))
    return out


def inputs():
    d = np.load(os.path.join(HERE, "brain1.npz"))
    rng = np.random.default_rng(2024)
    seeded = rng.integers(-300, 1200, size=(32, 30, 28)).astype(np.int16)
    return {"brain1": (d["image"], tuple(float(s) for s in d["spacing"])), "seeded": (seeded, (0.8, 1.0, 2.5))}


def main():
    try:
        import pywt      # noqa: F401
        import SimpleITK  # noqa: F401
    except ImportError as e:
        sys.exit("needs PyWavelets and SimpleITK (%s); nothing written" % e)
    out = {}
    for name, (arr, spacing) in inputs().items():
        out["%s__input" % name] = arr
        out["%s__spacing" % name] = np.array(spacing)
        for k, v in wavelet_subbands(arr, "coif1", 1).items():
            out["%s__wavelet_coif1_%s" % (name, k)] = v
        for k, v in log_images(arr, spacing, (1.0, 2.0, 3.0, 5.0)).items():
            out["%s__log_%s" % (name, k)] = v
    for k, v in wavelet_subbands(inputs()["seeded"][0], "db2", 2).items():
        out["seeded__wavelet_db2_%s" % k] = v
    path = os.path.join(HERE, "filters_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%d arrays)" % (path, len(out)))


if __name__ == "__main__":
    main()
