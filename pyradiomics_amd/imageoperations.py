"""Host-side pieces of the reference's `radiomics.imageoperations` that sit directly on the texture path:
grey-level discretisation (getBinEdges / binImage, imageoperations.py:67-174) and the ROI crop used before the
feature classes run (cropToTumorMask, imageoperations.py:407-445).  The wavelet / LoG filter stack lives in
pyradiomics_amd.filters.  One numpy pass each; not part of the measured kernel path."""
from __future__ import annotations

import logging

import numpy as np

from .image import Image, as_array

logger = logging.getLogger(__name__)


def getBinEdges(parameterValues, **kwargs):
    """Bin edges for np.digitize (half-open bins, 1-based levels).

    Fixed width (default binWidth=25): edges are multiples of the width anchored at 0, the lowest edge <= min(X),
    with one extra edge beyond max(X) so that the maximum gets its own half-open bin (imageoperations.py:128-139);
    a flat region that produces a single edge gets one bin of width 1 around it (:145-149).
    Fixed count (binCount): np.histogram edges with the last edge moved up by 1 (:122-126)."""
    binWidth = kwargs.get("binWidth", 25)
    binCount = kwargs.get("binCount")
    values = np.asarray(parameterValues)
    if binCount is not None:
        edges = np.histogram(values, binCount)[1]
        edges[-1] += 1
        return edges
    lo = min(values)
    hi = max(values)
    first = lo - (lo % binWidth)
    edges = np.arange(first, hi + 2 * binWidth, binWidth)
    if len(edges) == 1:
        edges = [edges[0] - 0.5, edges[0] + 0.5]
    return edges


def binImage(parameterMatrix, parameterMatrixCoordinates=None, **kwargs):
    """Discretises the ROI voxels of `parameterMatrix`; voxels outside the ROI become 0 (imageoperations.py:156-174).
    `parameterMatrixCoordinates` is a boolean mask or an index tuple.  Returns (levels int array, edges)."""
    matrix = np.asarray(parameterMatrix)
    if parameterMatrixCoordinates is None:
        edges = getBinEdges(matrix.flatten(), **kwargs)
        return np.digitize(matrix, edges), edges
    out = np.zeros(matrix.shape, dtype="int")
    roi = matrix[parameterMatrixCoordinates]
    edges = getBinEdges(roi, **kwargs)
    out[parameterMatrixCoordinates] = np.digitize(roi, edges)
    return out, edges


def boundingBox(maskArray):
    """(lo, hi) inclusive index bounds of the True voxels, numpy (z, y, x) order."""
    idx = np.where(maskArray)
    if len(idx[0]) == 0:
        raise ValueError("No labels found in this mask (i.e. nothing is segmented)!")
    lo = np.array([i.min() for i in idx])
    hi = np.array([i.max() for i in idx])
    return lo, hi


def cropToTumorMask(image, mask, label=1, padDistance=0):
    """Crops image and mask to the ROI bounding box padded by `padDistance` voxels, clipped to the image
    (imageoperations.py:407-445).  Accepts / returns pyradiomics_amd.image.Image."""
    img = image if isinstance(image, Image) else Image(as_array(image))
    msk = mask if isinstance(mask, Image) else Image(as_array(mask))
    m = msk.array == label
    lo, hi = boundingBox(m)
    lo = np.maximum(lo - padDistance, 0)
    hi = np.minimum(hi + padDistance, np.array(m.shape) - 1)
    sl = tuple(slice(int(a), int(b) + 1) for a, b in zip(lo, hi))
    nd = img.array.ndim
    d = np.array(img.direction, dtype=float).reshape(nd, nd)
    shift = d @ (np.array(img.spacing) * lo[::-1])
    origin = tuple(np.array(img.origin) + shift)
    return (Image(img.array[sl], img.spacing, origin, img.direction),
            Image(msk.array[sl], msk.spacing, origin, msk.direction))


def resegmentMask(image, mask, **kwargs):
    """Restricts the ROI to voxels whose intensity lies in `resegmentRange` (1 threshold: >= T; 2: closed range),
    with the thresholds absolute, relative to the ROI maximum, or in standard deviations around the ROI mean
    (imageoperations.py:533-640).  Returns a new mask Image holding `label` inside the kept ROI."""
    rng = kwargs["resegmentRange"]
    mode = kwargs.get("resegmentMode", "absolute")
    label = kwargs.get("label", 1)
    if rng is None:
        raise ValueError("resegmentRange is None.")
    if len(rng) == 0 or len(rng) > 2:
        raise ValueError("Length %d is not allowed for resegmentRange" % len(rng))
    im = as_array(image)
    roi = as_array(mask) == label
    if mode == "absolute":
        thr = sorted(rng)
    elif mode == "relative":
        top = np.max(im[roi])
        thr = [top * t for t in sorted(rng)]
    elif mode == "sigma":
        mu, sd = np.mean(im[roi]), np.std(im[roi])
        thr = [mu + sd * t for t in sorted(rng)]
    else:
        raise ValueError("Resegment mode %s not recognized." % mode)
    roi[roi] = im[roi] >= thr[0]
    if len(thr) == 2:
        roi[roi] = im[roi] <= thr[1]
    if np.sum(roi) <= 1:
        raise ValueError("Resegmentation excluded too many voxels with label %s (retained %d voxel(s))! "
                         "Cannot extract features" % (label, np.sum(roi)))
    out = np.zeros(roi.shape, dtype="int")
    out[roi] = label
    return mask.like(out) if isinstance(mask, Image) else Image(out)
