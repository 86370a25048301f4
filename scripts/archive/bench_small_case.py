#!/usr/bin/env python
"""A typical small clinical case (brain1: 256 x 256 x 25 image, 4137-voxel ROI, Original image, six classes): wall time
per execute() from files and from memory, device-resident route.  Small ROIs are launch- and Python-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
from pyradiomics_amd.image import read_nrrd
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
img, lbl = os.path.join(G, "brain1_image.nrrd"), os.path.join(G, "brain1_label.nrrd")
for label, params in (("Original", {"setting": {"binWidth": 25}}),
                      ("Original + Wavelet + LoG[2,3]", {"setting": {"binWidth": 25}, "imageType": {"Original": {}, "Wavelet": {}, "LoG": {"sigma": [2.0, 3.0]}}})):
    ex = RadiomicsFeatureExtractor(params)
    ex.execute(img, lbl)
    for src, a, b in (("files", img, lbl), ("memory", read_nrrd(img), read_nrrd(lbl))):
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t = time.perf_counter()
            r = ex.execute(a, b)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        print("%-32s from %-6s: %6.1f ms per case (%d values)" % (label, src, np.median(ts) * 1e3, len(r)), flush=True)
