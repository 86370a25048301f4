"""Host-side pieces of the reference's `radiomics.imageoperations` that sit directly on the texture path:
grey-level discretisation (getBinEdges / binImage, imageoperations.py:67-174) and the ROI crop used before the
feature classes run (cropToTumorMask, imageoperations.py:407-445).  The wavelet / LoG filter stack lives in
pyradiomics_amd.filters.  One numpy pass each; not part of the measured kernel path."""
from __future__ import annotations

import sys as _sys

import logging

import numpy as np

from .image import Image, as_array, as_image

logger = logging.getLogger(__name__)


def _engine():
    """the device engine (imports torch), loaded on first use: a function-level `from . import engine` goes through
    importlib's locked lookup on every call (11 us each, 54 of them per 256^3 case); sys.modules is a dict lookup"""
    m = _sys.modules.get("pyradiomics_amd.engine")
    if m is None:
        from . import engine as m
    return m


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
    """(lo, hi) inclusive index bounds of the True voxels, numpy (z, y, x) order.  Accepts a numpy array or a
    device tensor (reduced on the device; only 2*Nd integers come back)."""
    nd = maskArray.ndim
    lo, hi = np.zeros(nd, dtype=int), np.zeros(nd, dtype=int)
    for d in range(nd):
        other = tuple(k for k in range(nd) if k != d)
        if hasattr(maskArray, "data_ptr"):
            line = (maskArray.any(dim=other) if other else maskArray.ne(0)).cpu().numpy()
        else:
            line = maskArray.any(axis=other) if other else maskArray != 0
        idx = np.flatnonzero(line)
        if len(idx) == 0:
            raise ValueError("No labels found in this mask (i.e. nothing is segmented)!")
        lo[d], hi[d] = idx[0], idx[-1]
    return lo, hi


def roiTensor(mask, label=1):
    """boolean ROI (mask == label) of a mask Image as a device tensor, memoised on the Image"""
    key = ("roi", label)
    if key not in mask._derived:
        mask._derived[key] = (mask.device_tensor() == label)
    return mask._derived[key]


def cropToTumorMask(image, mask, label=1, padDistance=0, deviceResident=False, alignRows=True):
    """Crops image and mask to the ROI bounding box padded by `padDistance` voxels, clipped to the image
    (imageoperations.py:407-445).  Accepts / returns pyradiomics_amd.image.Image.  With `deviceResident` the crop
    is a device-to-device copy of the sub-box and the cropped mask Image is memoised on `mask`, so every derived
    image of one case shares one cropped ROI.  `alignRows=False` keeps the device crop at the reference's exact extent
    (the preCrop stage: filters applied afterwards see the crop's borders, so its extent must match the reference's)."""
    img, msk = as_image(image), as_image(mask)
    if deviceResident:
        # every derived image of a case is cropped with the same box: slices, origin shift and the cropped mask are worked out
        # once per (mask, box request) and per image geometry (0.1 ms of numpy per derived image otherwise)
        gkey = ("cropgeo", label, padDistance, bool(alignRows), img.spacing, img.origin, img.direction)
        geo = msk._derived.get(gkey)
        if geo is not None:
            sl, origin, cmask = geo
            return Image(None, img.spacing, origin, img.direction, tensor=img.device_tensor()[sl].contiguous()), cmask
    bkey = ("bbox", label)
    if bkey not in msk._derived:
        msk._derived[bkey] = boundingBox(roiTensor(msk, label) if deviceResident else (msk.array == label))
    lo, hi = msk._derived[bkey]
    lo = np.maximum(lo - padDistance, 0)
    hi = np.minimum(hi + padDistance, np.array(msk.shape) - 1)
    if deviceResident and alignRows:
        # Rows of a multiple of 4 voxels keep every kernel on its packed 4-voxels-per-lane path (a 231-wide crop of a
        # 256^3 case sent GLSZM, GLDM and NGTDM down their one-voxel-per-lane kernels: 2-3x slower).  The extra columns
        # lie outside the ROI's bounding box, so they are outside the ROI: no matrix, no statistic sees them.
        need = (-(int(hi[-1]) - int(lo[-1]) + 1)) % 4
        grow = min(need, int(msk.shape[-1]) - 1 - int(hi[-1]))
        hi = hi.copy()
        lo = lo.copy()
        hi[-1] += grow
        lo[-1] -= min(need - grow, int(lo[-1]))
    sl = tuple(slice(int(a), int(b) + 1) for a, b in zip(lo, hi))
    nd = len(img.shape)
    d = np.array(img.direction, dtype=float).reshape(nd, nd)
    shift = d @ (np.array(img.spacing) * lo[::-1])
    origin = tuple(np.array(img.origin) + shift)
    if not deviceResident:
        return (Image(img.array[sl], img.spacing, origin, img.direction),
                Image(msk.array[sl], msk.spacing, origin, msk.direction))
    ckey = ("crop", label, padDistance, bool(alignRows))
    if ckey not in msk._derived:
        msk._derived[ckey] = Image(None, msk.spacing, origin, msk.direction,
                                   tensor=msk.device_tensor()[sl].contiguous())
    msk._derived[gkey] = (sl, origin, msk._derived[ckey])
    return (Image(None, img.spacing, origin, img.direction, tensor=img.device_tensor()[sl].contiguous()),
            msk._derived[ckey])


# ---- resampling (imageoperations.py:448-612 on SimpleITK's ResampleImageFilter) ------------------------------------
# The arithmetic lives in ITK: BSplineDecompositionImageFilter (recursive prefilter, pole sqrt(3) - 2, mirror boundaries,
# 1e-10 horizon), BSplineInterpolateImageFunction (cubic weights, mirrored support), ResampleImageFilter (inside-buffer
# test, clamp + C cast to the input pixel type), NearestNeighborInterpolateImageFunction (round half up) for the
# mask.  Restated here from the published algorithms and PINNED by the reference's `_resampling` golden vectors
# (tests/golden/baseline_features.json: every feature of all six classes reproduces within 1e-12).
_BSPLINE_POLE = np.sqrt(3.0) - 2.0


def _bspline_decompose_axis(c, axis, tol=1e-10):
    """in-place-style 1-D cubic B-spline coefficient filter along `axis` (BSplineDecompositionImageFilter::
    DataToCoefficients1D with SetInitialCausalCoefficient / SetInitialAntiCausalCoefficient)"""
    c = np.moveaxis(c, axis, 0).copy()
    N = c.shape[0]
    if N == 1:
        return np.moveaxis(c, 0, axis)
    z = _BSPLINE_POLE
    c *= (1.0 - z) * (1.0 - 1.0 / z)
    horizon = int(np.ceil(np.log(tol) / np.log(abs(z))))
    if horizon < N:
        zn, acc = z, c[0].copy()
        for n in range(1, horizon):
            acc += zn * c[n]
            zn *= z
        c[0] = acc
    else:
        iz, z2n, zn = 1.0 / z, z ** (N - 1), z
        acc = c[0] + z2n * c[N - 1]
        z2n *= z2n * iz
        for n in range(1, N - 1):
            acc += (zn + z2n) * c[n]
            zn *= z
            z2n *= iz
        c[0] = acc / (1.0 - zn * zn)
    for n in range(1, N):
        c[n] += z * c[n - 1]
    c[N - 1] = (z / (z * z - 1.0)) * (z * c[N - 2] + c[N - 1])
    for n in range(N - 2, -1, -1):
        c[n] = z * (c[n + 1] - c[n])
    return np.moveaxis(c, 0, axis)


def _bspline_interp_axis(c, axis, pos):
    """cubic B-spline evaluation of the coefficient array along `axis` at the continuous indices `pos`"""
    N = c.shape[axis]
    base = np.floor(pos).astype(np.int64) - 1
    w = pos - (base + 1)
    w3 = (1.0 / 6.0) * w * w * w
    w0 = (1.0 / 6.0) + 0.5 * w * (w - 1.0) - w3
    w2 = w + w0 - 2.0 * w3
    w1 = 1.0 - w0 - w2 - w3
    out = 0
    shape = [1] * c.ndim
    shape[axis] = len(pos)
    for k, wk in enumerate((w0, w1, w2, w3)):
        idx = base + k
        if N == 1:
            idx = np.zeros_like(idx)
        else:                                               # mirror boundary conditions, period 2N - 2
            L2 = 2 * N - 2
            idx = np.where(idx < 0, -idx - L2 * ((-idx) // L2), idx - L2 * (idx // L2))
            idx = np.where(idx >= N, L2 - idx, idx)
        out = out + np.take(c, idx, axis=axis) * wk.reshape(shape)
    return out


def resampleImage(image, mask, **kwargs):
    """Resamples image (cubic B-spline; `interpolator: sitkNearestNeighbor | sitkLinear` also understood) and mask
    (nearest neighbour) onto the grid of spacing `resampledPixelSpacing` aligned to the input origin, restricted to the
    ROI bounding box + `padDistance` (imageoperations.py:448-612).  Image and mask must share one grid (the reference
    additionally accepts differing geometries through ITK's physical-space transforms)."""
    img, msk = as_image(image), as_image(mask)
    if img.shape != msk.shape or not np.allclose(img.spacing, msk.spacing) or not np.allclose(img.origin, msk.origin):
        raise NotImplementedError("resampleImage needs image and mask on the same grid")
    new = np.array(kwargs["resampledPixelSpacing"], dtype=float)
    interpolator = kwargs.get("interpolator", "sitkBSpline")
    pad = kwargs.get("padDistance", 5)
    label = int(kwargs.get("label", 1))
    old = np.array(msk.spacing)
    nd = len(old)
    if len(new) != nd:
        raise AssertionError("Wrong dimensionality (%d-D) of resampledPixelSpacing!, %d-D required" % (len(new), nd))
    new = np.where(new == 0, old, new)
    lo, hi = boundingBox(msk.array == label)
    lo, size = lo[::-1], (hi - lo + 1)[::-1]                # (x, y, z) like SimpleITK
    new = np.where(size != 1, new, old)
    if np.allclose(old, new):                               # nothing to interpolate: plain crop (:527-546)
        return cropToTumorMask(img, msk, label)
    ratio = old / new
    fullsize = np.array(msk.GetSize())
    L = np.floor((lo - 0.5) * ratio - pad)
    U = np.ceil((lo + size - 0.5) * ratio + pad)
    maxU = np.ceil(fullsize * ratio) - 1
    L = np.where(L < 0, 0, L)
    U = np.where(U > maxU, maxU, U)
    newsize = np.array(U - L + 1, dtype=int)
    start = 0.5 * (new - old) / old + L / ratio             # continuous input index of output voxel 0
    pos = [start[d] + np.arange(newsize[d]) * (new[d] / old[d]) for d in range(nd)]
    inside = None
    for d in range(nd):                                     # ImageFunction::IsInsideBuffer on the continuous index
        ok = (pos[d] >= -0.5) & (pos[d] < fullsize[d] - 0.5)
        shape = [1] * nd
        shape[nd - 1 - d] = len(ok)
        inside = ok.reshape(shape) if inside is None else inside & ok.reshape(shape)
    dirm = np.array(msk.direction, dtype=float).reshape(nd, nd)
    origin = tuple(np.array(msk.origin) + dirm @ (old * start))
    codes = {"sitkNearestNeighbor": 0, "1": 0, "sitkLinear": 1, "2": 1, "sitkBSpline": 3, "3": 3}
    if str(interpolator) not in codes:
        raise NotImplementedError("interpolator %r" % (interpolator,))
    if (kwargs.get("deviceResident", False) or img.on_device) and nd <= 3:
        # same arithmetic in the same order on the device (prad_resample_dev): bit-identical to the numpy route below
        engine = _engine()
        step = (new / old)[::-1]
        ri = engine.resample(img.device_tensor(), start[::-1], step, newsize[::-1], codes[str(interpolator)])
        rm = engine.resample(msk.device_tensor(), start[::-1], step, newsize[::-1], 0)
        # The upload widens narrow integer pixel types (uint8 -> int16, uint16 -> int32 ...), and the kernel clamps to the
        # range of what it was given; ITK's CastPixelWithBoundsChecking (and the host route below) clamp to the ORIGINAL
        # pixel type: B-spline overshoot of an unsigned image must not survive as negative / above-maximum values.
        src_dtype = img._array.dtype if getattr(img, "_array", None) is not None else None
        if src_dtype is not None and np.issubdtype(src_dtype, np.integer) and str(interpolator) in ("sitkBSpline", "3"):
            info = np.iinfo(src_dtype)
            ri = ri.clamp(min=int(info.min), max=int(info.max))
        return (Image(None, tuple(new), origin, msk.direction, tensor=ri),
                Image(None, tuple(new), origin, msk.direction, tensor=rm))
    src = img.array
    if str(interpolator) in ("sitkBSpline", "3"):
        c = src.astype(np.float64)
        for d in range(nd):                                 # ITK filters dimension 0 (x) first
            c = _bspline_decompose_axis(c, nd - 1 - d)
        val = c
        for d in range(nd):
            val = _bspline_interp_axis(val, nd - 1 - d, pos[d])
    elif str(interpolator) in ("sitkLinear", "2"):
        val = src.astype(np.float64)
        for d in range(nd):
            ax, N = nd - 1 - d, src.shape[nd - 1 - d]
            b = np.floor(pos[d]).astype(np.int64)
            f = pos[d] - b
            shape = [1] * nd
            shape[ax] = len(f)
            val = (np.take(val, np.clip(b, 0, N - 1), axis=ax) * (1 - f).reshape(shape)
                   + np.take(val, np.clip(b + 1, 0, N - 1), axis=ax) * f.reshape(shape))
    elif str(interpolator) in ("sitkNearestNeighbor", "1"):
        val = src[np.ix_(*[np.clip(np.floor(pos[d] + 0.5).astype(int), 0, fullsize[d] - 1) for d in range(nd - 1, -1, -1)])]
    else:
        raise NotImplementedError("interpolator %r" % (interpolator,))
    if np.issubdtype(src.dtype, np.integer):                # CastPixelWithBoundsChecking: clamp, then a C cast
        info = np.iinfo(src.dtype)
        val = np.trunc(np.clip(val, info.min, info.max))
    out = np.where(inside, val, 0).astype(src.dtype)
    near = [np.clip(np.floor(pos[d] + 0.5).astype(int), 0, fullsize[d] - 1) for d in range(nd)]
    m = np.where(inside, msk.array[np.ix_(*near[::-1])], 0).astype(msk.array.dtype)
    return Image(out, tuple(new), origin, msk.direction), Image(m, tuple(new), origin, msk.direction)


def normalizeImage(image, **kwargs):
    """f(x) = scale * (x - mean) / sigma over ALL voxels of the image (imageoperations.py:615-654: sitk.Normalize, i.e.
    float64 output and the N - 1 standard deviation -- pinned by the reference's `_normalization` golden vectors),
    values beyond +-removeOutliers standard deviations clipped before the scale is applied.  Works on the device when
    the image lives there (or `deviceResident` is set), on the host otherwise."""
    scale = kwargs.get("normalizeScale", 1)
    outliers = kwargs.get("removeOutliers")
    img = as_image(image)
    if img.on_device or kwargs.get("deviceResident", False):
        import torch
        x = img.device_tensor().to(torch.float64)
        mean = x.mean()
        sigma = torch.sqrt(((x - mean) ** 2).sum() / (x.numel() - 1))
        out = (x - mean) / sigma
        if outliers is not None:
            out = out.clamp(-outliers, outliers)
        return img.like(tensor=out * float(scale))
    x = img.array.astype(np.float64)
    mean = x.mean()
    sigma = np.sqrt(((x - mean) ** 2).sum() / (x.size - 1))
    out = (x - mean) / sigma
    if outliers is not None:
        out = np.clip(out, -outliers, outliers)
    return img.like(out * float(scale))


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


# ---- image / mask geometry (imageoperations.py:177-402: checkMask step 1, _correctMask, _checkROI) -----------------
def _index_to_physical(img):
    """(origin [Nd], matrix [Nd, Nd]) with physical(x, y, z) = origin + matrix @ index(x, y, z)"""
    nd = len(img.shape)
    D = np.asarray(img.GetDirection(), dtype=np.float64).reshape(nd, nd)
    return np.asarray(img.GetOrigin(), dtype=np.float64), D * np.asarray(img.GetSpacing(), dtype=np.float64)[None, :]


def sameGeometry(image, mask, tolerance=None):
    """ITK's ImageToImageFilter::VerifyInputInformation, the test sitk.LabelStatisticsImageFilter applies before it
    looks at a voxel: equal sizes, and origin / spacing within `tolerance` x spacing[0] of the image, direction within
    `tolerance` (element-wise; default 1e-6, the ITK default the reference leaves in place unless ``geometryTolerance``
    is set, featureextractor.py:114-126).  Returns (ok, reason)."""
    tol = 1e-6 if tolerance is None else float(tolerance)
    if image.shape != mask.shape:
        return False, "size"
    ctol = abs(tol * image.GetSpacing()[0])
    for name, a, b, t in (("origin", image.GetOrigin(), mask.GetOrigin(), ctol),
                          ("spacing", image.GetSpacing(), mask.GetSpacing(), ctol),
                          ("direction", image.GetDirection(), mask.GetDirection(), tol)):
        if np.any(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) > t):
            return False, name
    return True, ""


def _checkROI(image, mask, **kwargs):
    """imageoperations.py:341-402: the label must be present and the physical box of its ROI (voxel edges, hence the
    half voxel) must lie inside the image's, with the reference's 1e-3 voxel tolerance"""
    label = int(kwargs.get("label", 1))
    roi = mask.array == label
    if not roi.any():
        raise ValueError("Label (%d) not present in mask" % label)
    lo, hi = boundingBox(roi)                                  # numpy (z, y, x) order, inclusive
    lo_xyz, hi_xyz = lo[::-1].astype(np.float64), hi[::-1].astype(np.float64)
    mo, mm = _index_to_physical(mask)
    io, im = _index_to_physical(image)
    inv = np.linalg.inv(im)
    bounds = np.array([inv @ (mo + mm @ (lo_xyz - 0.5) - io), inv @ (mo + mm @ (hi_xyz + 0.5) - io)])
    tol = 1e-3
    size_xyz = np.asarray(image.shape[::-1], dtype=np.float64)
    if np.any(bounds.min(axis=0) < -0.5 - tol) or np.any(bounds.max(axis=0) > size_xyz - 0.5 + tol):
        raise ValueError("Bounding box of ROI is larger than image space:\n\tROI bounds (x, y, z image coordinate space) "
                         "%s\n\tImage Size %s" % (bounds.tolist(), tuple(image.shape[::-1])))


def _correctMask(image, mask, **kwargs):
    """imageoperations.py:315-338: the mask resampled onto the image grid, nearest neighbour (ITK rounds half-way
    continuous indices up), zero outside the mask's buffer"""
    _checkROI(image, mask, **kwargs)
    nd = len(image.shape)
    mo, mm = _index_to_physical(mask)
    io, im = _index_to_physical(image)
    A = np.linalg.inv(mm) @ im                                 # image index (x, y, z) -> mask continuous index
    b = np.linalg.inv(mm) @ (io - mo)
    src = mask.array
    out = np.zeros(image.shape, dtype=src.dtype)
    axes = [np.arange(n, dtype=np.float64) for n in image.shape[::-1]]      # x, y, z index ranges
    if np.allclose(A, np.diag(np.diag(A)), rtol=0, atol=1e-12):
        # axis-aligned grids (the usual case): one index vector per axis
        idx = []
        for d in range(nd):
            c = np.floor(A[d, d] * axes[d] + b[d] + 0.5).astype(np.int64)
            idx.append(c)
        ok = [(c >= 0) & (c < src.shape[nd - 1 - d]) for d, c in enumerate(idx)]
        sel = np.ix_(*[np.nonzero(ok[d])[0] for d in range(nd - 1, -1, -1)])            # output (z, y, x)
        pick = np.ix_(*[idx[d][ok[d]] for d in range(nd - 1, -1, -1)])                  # source (z, y, x)
        out[sel] = src[pick]
    else:
        grid = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, nd)     # rows (x, y, z), x slowest
        cont = np.floor(grid @ A.T + b + 0.5).astype(np.int64)
        inside = np.all((cont >= 0) & (cont < np.asarray(src.shape[::-1])), axis=1)
        vals = np.zeros(len(grid), dtype=src.dtype)
        vals[inside] = src[tuple(cont[inside, d] for d in range(nd - 1, -1, -1))]
        out = np.ascontiguousarray(vals.reshape(image.shape[::-1]).transpose(tuple(range(nd - 1, -1, -1))))
    return image.like(out)


def checkMaskGeometry(image, mask, **kwargs):
    """Step 1 of the reference's checkMask (imageoperations.py:241-287): a mask that does not share the image's grid
    is an error, unless ``correctMask`` asks for it to be resampled onto the image grid.  Returns the mask to use."""
    ok, reason = sameGeometry(image, mask, kwargs.get("geometryTolerance"))
    if ok:
        return mask
    if not kwargs.get("correctMask", False):
        if reason == "size":
            raise ValueError("Image/Mask datatype or size mismatch. Potential fix: enable correctMask, see "
                             "Documentation:Usage:Customizing the Extraction:Settings:correctMask for more information")
        raise ValueError("Image/Mask geometry mismatch. Potential fix: increase tolerance using geometryTolerance, "
                         "see Documentation:Usage:Customizing the Extraction:Settings:geometryTolerance for more "
                         "information")
    logger.warning("Image/Mask geometry mismatch, attempting to correct Mask")
    return _correctMask(image, mask, **kwargs)


def getMask(mask, **kwargs):
    """imageoperations.py:12-64: picks channel ``label_channel`` (default 0) of a segmentation stored as a vector image
    (array (z, y, x, c): an NRRD with a component axis, or an array with one more axis than its geometry) and checks
    that the label occurs"""
    label = kwargs.get("label", 1)
    channel = int(kwargs.get("label_channel", 0) or 0)
    ncomp = getattr(mask, "components", None)
    if ncomp is None and len(mask.shape) == len(mask.GetSpacing()) + 1:
        ncomp = mask.shape[-1]
    if ncomp is not None:
        if not 0 <= channel < ncomp:
            raise ValueError("Mask %d requested, but segmentation object only contains %d objects" % (channel, ncomp))
        logger.info("Extracting mask at index %d", channel)
        mask = Image(np.ascontiguousarray(mask.array[..., channel]), mask.GetSpacing(), mask.GetOrigin(),
                     mask.GetDirection())
    arr = mask.array
    if not np.any(arr == label):
        labels = np.unique(arr)
        if len(labels) == 1 and labels[0] == 0:
            raise ValueError("No labels found in this mask (i.e. nothing is segmented)!")
        raise ValueError("Label (%g) not present in mask. Choose from %s" % (label, labels[labels != 0]))
    return mask


def checkMask(image, mask, **kwargs):
    """The reference's entry point (imageoperations.py:177-312) for callers that use it directly: geometry check /
    correction, label present, ``minimumROIDimensions`` and ``minimumROISize``.  Returns (boundingBox, correctedMask)
    with boundingBox = (L_x, U_x, L_y, U_y, L_z, U_z) as the reference, correctedMask None unless the mask was resampled.
    (RadiomicsFeatureExtractor.execute performs the same checks on the device-resident case.)"""
    label = int(kwargs.get("label", 1))
    checked = checkMaskGeometry(image, mask, **kwargs)
    corrected = None if checked is mask else checked
    roi = checked.array == label
    if not roi.any():
        raise ValueError("Label (%g) not present in mask" % label)
    lo, hi = boundingBox(roi)
    bb = np.empty(2 * len(lo), dtype=np.int64)
    bb[0::2], bb[1::2] = lo[::-1], hi[::-1]
    ndims = int(np.sum(hi - lo + 1 > 1))
    if ndims == 0:
        raise ValueError("mask only contains 1 segmented voxel! Cannot extract features for a single voxel.")
    minDims = kwargs.get("minimumROIDimensions", 2)
    if ndims < minDims:
        raise ValueError("mask has too few dimensions (number of dimensions %d, minimum required %d)" % (ndims, minDims))
    minSize = kwargs.get("minimumROISize")
    if minSize is not None and int(roi.sum()) <= minSize:
        raise ValueError("Size of the ROI is too small (minimum size: %g, ROI size: %g" % (minSize, int(roi.sum())))
    return bb, corrected
