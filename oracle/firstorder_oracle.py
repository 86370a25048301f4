"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's first-order statistics,
radiomics/firstorder.py:33-474, used by tests/ (and by the CPU operator backend oracle.binding.CMatricesCPU) to check
the MI355X path (prad_firstorder_dev / prad_voxel_firstorder_dev).  Pinned by the reference's golden vectors
data/baseline/baseline_firstorder.csv (tests/golden/baseline_features.json, tests/test_firstorder.py).

The reference applies nan-aware numpy reductions to `targetVoxelArray`: shape (1, Np) in segment mode
(firstorder.py:96-101), (Nvox, Nk) gathered from a NaN-padded, NaN-outside-ROI copy of the image at
centre + kernelOffsets in voxel mode (:37-94, :104-118).  Both are restated here on the same array layout."""
import numpy as np

FIELDS = ("Np", "Energy", "Minimum", "P10", "P25", "Median", "P75", "P90", "Maximum", "Mean", "MAD", "rMAD",
          "m2", "m3", "m4")


def _stats_rows(target, shift):
    """per-row statistics of a (rows, Nk) float array with NaN = absent (firstorder.py:136-462)"""
    x = np.asarray(target, dtype=np.float64)
    with np.errstate(all="ignore"):
        n = np.sum(~np.isnan(x), 1).astype(float)
        mean = np.nanmean(x, 1)
        d = x - mean[:, None]
        p10 = np.nanpercentile(x, 10, axis=1)
        p90 = np.nanpercentile(x, 90, axis=1)
        band = x.copy()                                     # :323-342 robust mean absolute deviation
        ok = ~np.isnan(band)
        ok[ok] = ((band - p10[:, None])[ok] < 0) | ((band - p90[:, None])[ok] > 0)
        band[ok] = np.nan
        rmad = np.nanmean(np.absolute(band - np.nanmean(band, 1, keepdims=True)), 1)
        return {"Np": n, "Energy": np.nansum((x + shift) ** 2, 1), "Minimum": np.nanmin(x, 1), "P10": p10,
                "P25": np.nanpercentile(x, 25, axis=1), "Median": np.nanmedian(x, 1),
                "P75": np.nanpercentile(x, 75, axis=1), "P90": p90, "Maximum": np.nanmax(x, 1), "Mean": mean,
                "MAD": np.nanmean(np.absolute(d), 1), "rMAD": rmad, "m2": np.nanmean(d ** 2, 1),
                "m3": np.nanmean(d ** 3, 1), "m4": np.nanmean(d ** 4, 1)}


def firstorder_stats(image, mask, voxelArrayShift=0.0):
    """segment mode: {field: float} over image[mask]"""
    vals = np.asarray(image)[np.asarray(mask).astype(bool)].astype(float).reshape(1, -1)
    return {k: float(v[0]) for k, v in _stats_rows(vals, voxelArrayShift).items()}


def kernel_offsets(bbsize, kernelRadius, force2D, force2Ddimension):
    """firstorder.py:45-67: all offsets of infinity norm 1..kernelRadius (generate_angles, bidirectional; components
    limited to |o| < bbsize[d], none along the force2D dimension) plus the centre.  -> int [Nk, Nd]"""
    nd = len(bbsize)
    half = [min(kernelRadius, max(int(b) - 1, 0)) for b in bbsize]
    if force2D:
        half[force2Ddimension] = 0
    grids = np.meshgrid(*[np.arange(-h, h + 1) for h in half], indexing="ij")
    return np.stack([g.ravel() for g in grids], 1).reshape(-1, nd)


def voxel_target_array(image, mask, voxels, kernelRadius, bbsize, force2D=False, force2Ddimension=0):
    """(Nvox, Nk) intensities of every kernel, NaN where the kernel leaves the image or the mask (:69-94,:104-112)"""
    img = np.asarray(image).astype(float)
    img[~np.asarray(mask).astype(bool)] = np.nan
    img = np.pad(img, kernelRadius, mode="constant", constant_values=np.nan)
    off = kernel_offsets(bbsize, kernelRadius, force2D, force2Ddimension).T          # (Nd, Nk)
    coords = off[:, None, :] + (np.asarray(voxels) + kernelRadius)[:, :, None]       # (Nd, Nvox, Nk)
    return img[tuple(coords)], tuple(coords)


def voxel_firstorder(image, mask, levels, voxels, kernelRadius, bbsize, force2D, force2Ddimension, voxelArrayShift,
                     voxelVolume, features):
    """{feature name: float64 [Nvox]} with the reference's formulas on the gathered windows"""
    target, coords = voxel_target_array(image, mask, voxels, kernelRadius, bbsize, force2D, force2Ddimension)
    st = _stats_rows(target, voxelArrayShift)
    lev = np.pad(np.where(np.asarray(mask).astype(bool), np.asarray(levels), 0), kernelRadius)[coords]   # (Nvox, Nk)
    grays = np.unique(lev[lev > 0])
    p = np.stack([np.sum(lev == g, 1) for g in grays], 1).astype(float)                    # :113-118
    tot = np.sum(p, 1, keepdims=True)
    tot[tot == 0] = 1
    p /= tot
    return {f: derive(f, st, p, voxelVolume) for f in features}


def derive(name, st, p_i, voxelVolume=1.0):
    """feature value(s) from the statistics dict and the normalised level histogram p_i (rows, Ngp)"""
    eps = np.spacing(1)
    with np.errstate(all="ignore"):
        if name == "Energy":
            return st["Energy"]
        if name == "TotalEnergy":
            return st["Energy"] * voxelVolume
        if name == "Entropy":
            return -1.0 * np.sum(p_i * np.log2(p_i + eps), 1)
        if name == "Uniformity":
            return np.nansum(p_i ** 2, 1)
        if name in ("Minimum", "Maximum", "Mean", "Median"):
            return st[name]
        if name == "10Percentile":
            return st["P10"]
        if name == "90Percentile":
            return st["P90"]
        if name == "InterquartileRange":
            return st["P75"] - st["P25"]
        if name == "Range":
            return st["Maximum"] - st["Minimum"]
        if name == "MeanAbsoluteDeviation":
            return st["MAD"]
        if name == "RobustMeanAbsoluteDeviation":
            return st["rMAD"]
        if name == "RootMeanSquared":
            return np.sqrt(st["Energy"] / st["Np"])
        if name == "StandardDeviation":
            return np.sqrt(st["m2"])
        if name == "Variance":
            return np.sqrt(st["m2"]) ** 2
        m2 = np.where(np.asarray(st["m2"]) == 0, 1.0, st["m2"])            # :403-405, :441-443
        if name == "Skewness":
            return st["m3"] / m2 ** 1.5
        if name == "Kurtosis":
            return st["m4"] / m2 ** 2.0
    raise KeyError(name)
