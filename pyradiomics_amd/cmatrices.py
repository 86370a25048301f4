"""Drop-in for the reference's native operator module `radiomics._cmatrices`
(bound as `radiomics.cMatrices`, radiomics/__init__.py:343-349).

Same six functions, same positional signatures, same dtype coercions, output shapes/dtypes and exception
types as radiomics/src/_cmatrices.c -- but every matrix is built on the MI355X through the C ABI of
include/pyradiomics_amd.h (ctypes).  A reference checkout switches over with

    import radiomics, pyradiomics_amd.cmatrices
    radiomics.cMatrices = pyradiomics_amd.cmatrices          # the one line at radiomics/__init__.py:348

Inputs are numpy arrays (host) as in the reference.  In segment mode the same functions also accept torch tensors
that already live in HBM (`DEVICE_TENSORS`): nothing is staged through the host and only the finished matrix comes
back as numpy, which is how pyradiomics_amd's own feature classes call them (pyradiomics_amd/base.py).  The
lower-level device-resident entry points live in pyradiomics_amd.engine.
"""
from __future__ import annotations

import sys as _sys

import ctypes as C

import os

import numpy as np

from . import _lib

_ip = C.POINTER(C.c_int)

def _engine():
    """the device engine (imports torch), loaded on first use: a function-level `from . import engine` goes through
    importlib's locked lookup on every call (11 us each, 54 of them per 256^3 case); sys.modules is a dict lookup"""
    m = _sys.modules.get("pyradiomics_amd.engine")
    if m is None:
        from . import engine as m
    return m


DEVICE_TENSORS = True     # calculate_* accept device tensors in segment mode (checked by pyradiomics_amd.base)


def _on_device(x):
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


def _memo(image, mask):
    """per-levels-tensor memo used to serve GLCM and GLRLM from one fused sweep; only active on tensors that
    pyradiomics_amd.base tagged (`_prad_memo`), keyed by the mask they were tagged with"""
    memo = getattr(image, "_prad_memo", None)
    if memo is None or memo.get("mask") is not mask:
        return None
    return memo


class _PairsRuns(dict):
    """{"glcm_dev", "glrlm_dev", "angles"} + the host copies "glcm" / "glrlm" ([1, Ng, ., Na], the reference's layout),
    made when somebody asks (the fused segment route never does: a 256^3 GLRLM is 0.8 MB per derived image)"""

    def __missing__(self, key):
        if key in ("glcm", "glrlm"):
            dev = self[key + "_dev"]
            val = None if dev is None else dev.cpu().numpy()[None]
            self[key] = val
            return val
        raise KeyError(key)


def _dev_pairs_runs(image, mask, Ng, force2D, force2Ddimension, want, deferred=False):
    """distance-1 GLCM / GLRLM of device tensors through the fused sweep.  deferred=True: the kernels are only enqueued
    on the current stream (engine.glcm_glrlm(deferred=True) + deferred_join); engine.deferred_status() tells afterwards
    whether the levels were inside [1, Ng]"""
    engine = _engine()
    f2d = int(force2Ddimension) if force2D else -1
    Nr = int(max(image.shape))
    memo = _memo(image, mask)
    key = ("pairs_runs", int(Ng), f2d)
    if memo is not None and key in memo:
        return memo[key]
    both = memo is not None
    g, r, angles = engine.glcm_glrlm(image, mask, int(Ng), Nr, force2D, force2Ddimension,
                                     want_glcm=both or want == "glcm", want_glrlm=both or want == "glrlm",
                                     deferred=deferred)
    if deferred:
        engine.deferred_join()          # (pipeline mode: the walk of this volume is launched now, not by the next call)
    res = _PairsRuns({"angles": angles, "glcm_dev": g, "glrlm_dev": r})
    if memo is not None:
        memo[key] = res
    return res


def _dev_gldm_ngtdm(image, mask, Ng, alpha, dist, force2D, force2Ddimension, deferred):
    """(GLDM, NGTDM) device matrices of a segment.  On tensors pyradiomics_amd.base tagged (all feature classes of one
    derived image share them) both come from ONE pass over the neighbourhoods (engine.gldm_ngtdm) and are kept for the
    other class; alpha = None: the caller only wants NGTDM (the fused pass runs with the default alpha 0)"""
    engine = _engine()
    f2d = int(force2Ddimension) if force2D else -1
    memo = _memo(image, mask)
    a = 0 if alpha is None else int(alpha)
    key = ("gldm_ngtdm", int(Ng), f2d, tuple(dist), a)
    if memo is not None and not os.environ.get("PRAD_NO_FUSED_NEIGH"):
        for k, v in memo.items():          # NGTDM does not depend on alpha: any fused result of this geometry serves it
            if isinstance(k, tuple) and k[:4] == key[:4] and (alpha is None or k[4] == a):
                return v
        res = engine.gldm_ngtdm(image, mask, int(Ng), a, dist, force2D, force2Ddimension, deferred=deferred)
        memo[key] = res
        return res
    if alpha is None:
        return None, engine.ngtdm(image, mask, int(Ng), dist, force2D, force2Ddimension, deferred=deferred)
    return engine.gldm(image, mask, int(Ng), a, dist, force2D, force2Ddimension, deferred=deferred), None


def _iptr(a):
    return a.ctypes.data_as(_ip)


def _vptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


def _parse_arrays(image, mask):
    """_cmatrices.c:1023-1085 (try_parse_arrays): int32 / bool, C-contiguous, equal rank and shape."""
    img = np.ascontiguousarray(np.asarray(image).astype(np.intc, copy=False))
    msk = np.asarray(mask)
    if msk.dtype in (np.uint8, np.int8):
        msk = np.ascontiguousarray(msk).view(np.uint8)     # one byte per voxel already, the kernels test != 0: no copy
    else:
        msk = np.ascontiguousarray(msk.astype(np.bool_, copy=False))
    if img.ndim != msk.ndim:
        raise ValueError("Expected image and mask to have equal number of dimensions.")
    if img.shape != msk.shape:
        raise ValueError("Dimensions of image and mask do not match.")
    size = np.array(img.shape, dtype=np.intc)
    return img, msk, size


def _parse_voxels(voxels, Nd, kernelRadius):
    """_cmatrices.c:1087-1118 (try_parse_voxels_arr)."""
    if voxels is None:
        return None, 1
    if kernelRadius <= 0:
        raise RuntimeError("Expecting kernelRadius > 0")
    v = np.ascontiguousarray(np.asarray(voxels).astype(np.intc, copy=False))
    if v.ndim != 2 or v.shape[0] != Nd:
        raise RuntimeError("Expecting voxel indices array to be 2-dimensional")
    return v, int(v.shape[1])


_angle_memo: dict = {}


def _build_angles(size, distances, bidirectional, force2Ddim):
    """_cmatrices.c:926-1021 (build_angles_arr) on prad_get_angle_count / prad_build_angles."""
    lib = _lib.load()
    if distances is None:
        dist = np.array([1], dtype=np.intc)
    else:
        try:
            dist = np.ascontiguousarray(np.asarray(distances).astype(np.intc, copy=False))
        except (TypeError, ValueError):
            raise RuntimeError("Error parsing distances array.")
        if dist.ndim != 1:
            raise ValueError("Expecting distances array to be 1-dimensional.")
    Nd = int(size.shape[0])
    key = (size.tobytes(), dist.tobytes(), bool(bidirectional), int(force2Ddim))
    hit = _angle_memo.get(key)         # (27 calls per case with 5 distinct answers; the table is never written to)
    if hit is not None:
        return hit
    Na = lib.prad_get_angle_count(_iptr(size), _iptr(dist), Nd, int(dist.shape[0]), int(bool(bidirectional)),
                                  int(force2Ddim)) if dist.shape[0] > 0 else 0
    if Na == 0:
        raise RuntimeError("Error getting angle count.")
    angles = np.empty((Na, Nd), dtype=np.intc)
    if lib.prad_build_angles(_iptr(size), _iptr(dist), Nd, int(dist.shape[0]), int(force2Ddim), Na, _iptr(angles)) > 0:
        raise RuntimeError("Error building angles.")
    angles.setflags(write=False)
    if len(_angle_memo) > 256:
        _angle_memo.clear()
    _angle_memo[key] = angles
    return angles


def _common(image, mask, distances, bidirectional, force2D, force2Ddimension, kernelRadius, voxels):
    img, msk, size = _parse_arrays(image, mask)
    vox, Nvox = _parse_voxels(voxels, img.ndim, kernelRadius)
    f2d = int(force2Ddimension) if force2D else -1
    angles = _build_angles(size, distances, bidirectional, f2d)
    return img, msk, size, vox, Nvox, f2d, angles


def calculate_glcm(image, mask, distances, Ng, force2D, force2Ddimension, kernelRadius=0, voxels=None):
    """-> (P float64 [Nvox, Ng, Ng, Na], angles int32 [Na, Nd]);  _cmatrices.c:84-233"""
    if _on_device(image) and voxels is None:
        engine = _engine()
        dist = [int(d) for d in np.asarray(distances).ravel()]
        if dist == [1]:
            res = _dev_pairs_runs(image, mask, Ng, force2D, force2Ddimension, "glcm")
            return res["glcm"], res["angles"]
        P, angles = engine.glcm(image, mask, int(Ng), dist, force2D, force2Ddimension)
        return P.cpu().numpy()[None], angles
    img, msk, size, vox, Nvox, f2d, angles = _common(image, mask, distances, False, force2D, force2Ddimension,
                                                     kernelRadius, voxels)
    Na, Nd = angles.shape
    out = np.empty((Nvox, Ng, Ng, Na), dtype=np.float64)
    rc = _lib.load().prad_calculate_glcm(_vptr(img), _vptr(msk), _iptr(size), Nd, _iptr(angles), Na, int(Ng),
                                         Nvox, _vptr(vox), int(kernelRadius), f2d, _vptr(out))
    _lib.raise_for(rc, "GLCM")
    return out, angles


def calculate_glrlm(image, mask, Ng, Nr, force2D, force2Ddimension, kernelRadius=0, voxels=None):
    """-> (P float64 [Nvox, Ng, Nr, Na], angles);  _cmatrices.c:432-581 (distances fixed to [1])"""
    if _on_device(image) and voxels is None:
        if int(Nr) == int(max(image.shape)):
            res = _dev_pairs_runs(image, mask, Ng, force2D, force2Ddimension, "glrlm")
            return res["glrlm"], res["angles"]
        engine = _engine()
        _, r, angles = engine.glcm_glrlm(image, mask, int(Ng), int(Nr), force2D, force2Ddimension, want_glcm=False)
        return r.cpu().numpy()[None], angles
    img, msk, size, vox, Nvox, f2d, angles = _common(image, mask, None, False, force2D, force2Ddimension,
                                                     kernelRadius, voxels)
    Na, Nd = angles.shape
    out = np.empty((Nvox, Ng, Nr, Na), dtype=np.float64)
    rc = _lib.load().prad_calculate_glrlm(_vptr(img), _vptr(msk), _iptr(size), Nd, _iptr(angles), Na, int(Ng),
                                          int(Nr), Nvox, _vptr(vox), int(kernelRadius), f2d, _vptr(out))
    _lib.raise_for(rc, "GLRLM")
    return out, angles


def calculate_glcm_glrlm(image, mask, Ng, Nr, force2D, force2Ddimension, kernelRadius=0, voxels=None):
    """Both matrices for distance 1 from one pack + one sweep per angle (no reference analogue; results equal
    calculate_glcm(..., [1], ...) and calculate_glrlm(...)).  -> (P_glcm, P_glrlm, angles)"""
    img, msk, size, vox, Nvox, f2d, angles = _common(image, mask, None, False, force2D, force2Ddimension,
                                                     kernelRadius, voxels)
    Na, Nd = angles.shape
    glcm = np.empty((Nvox, Ng, Ng, Na), dtype=np.float64)
    glrlm = np.empty((Nvox, Ng, Nr, Na), dtype=np.float64)
    rc = _lib.load().prad_calculate_glcm_glrlm(_vptr(img), _vptr(msk), _iptr(size), Nd, _iptr(angles), Na, int(Ng),
                                               int(Nr), Nvox, _vptr(vox), int(kernelRadius), f2d, _vptr(glcm),
                                               _vptr(glrlm))
    _lib.raise_for(rc, "GLCM+GLRLM")
    return glcm, glrlm, angles


def calculate_gldm(image, mask, distances, Ng, alpha, force2D, force2Ddimension, kernelRadius=0, voxels=None):
    """-> P float64 [Nvox, Ng, 2*Na+1] (Na = bidirectional angle count);  _cmatrices.c:731-880"""
    if _on_device(image) and voxels is None:
        engine = _engine()
        return engine.gldm(image, mask, int(Ng), int(alpha), [int(d) for d in np.asarray(distances).ravel()], force2D,
                           force2Ddimension).cpu().numpy()[None]
    img, msk, size, vox, Nvox, f2d, angles = _common(image, mask, distances, True, force2D, force2Ddimension,
                                                     kernelRadius, voxels)
    Na, Nd = angles.shape
    out = np.empty((Nvox, Ng, 2 * Na + 1), dtype=np.float64)
    rc = _lib.load().prad_calculate_gldm(_vptr(img), _vptr(msk), _iptr(size), Nd, _iptr(angles), Na, int(Ng),
                                         int(alpha), Nvox, _vptr(vox), int(kernelRadius), f2d, _vptr(out))
    _lib.raise_for(rc, "GLDM")
    return out


def calculate_ngtdm(image, mask, distances, Ng, force2D, force2Ddimension, kernelRadius=0, voxels=None):
    """-> P float64 [Nvox, Ng, 3];  _cmatrices.c:583-729"""
    if _on_device(image) and voxels is None:
        engine = _engine()
        return engine.ngtdm(image, mask, int(Ng), [int(d) for d in np.asarray(distances).ravel()], force2D,
                            force2Ddimension).cpu().numpy()[None]
    img, msk, size, vox, Nvox, f2d, angles = _common(image, mask, distances, True, force2D, force2Ddimension,
                                                     kernelRadius, voxels)
    Na, Nd = angles.shape
    out = np.empty((Nvox, Ng, 3), dtype=np.float64)
    rc = _lib.load().prad_calculate_ngtdm(_vptr(img), _vptr(msk), _iptr(size), Nd, _iptr(angles), Na, int(Ng),
                                          Nvox, _vptr(vox), int(kernelRadius), f2d, _vptr(out))
    _lib.raise_for(rc, "NGTDM")
    return out


def calculate_glszm(image, mask, Ng, Ns, force2D, force2Ddimension, kernelRadius=0, voxels=None):
    """-> P float64 [Nvox, Ng, maxRegion], last axis cropped to the largest zone found (>= 1);
    _cmatrices.c:235-430.  The input mask is not modified (the reference works on a private copy)."""
    if _on_device(image) and voxels is None:
        engine = _engine()
        return engine.glszm(image, mask, int(Ng), int(Ns), force2D, force2Ddimension).cpu().numpy()[None]
    img, msk, size, vox, Nvox, f2d, angles = _common(image, mask, None, True, force2D, force2Ddimension,
                                                     kernelRadius, voxels)
    Na, Nd = angles.shape
    lib = _lib.load()
    nz = C.c_longlong(0)
    rc = lib.prad_calculate_glszm(_vptr(img), _vptr(msk), _iptr(size), Nd, _iptr(angles), Na, int(Ng), int(Ns),
                                  Nvox, _vptr(vox), int(kernelRadius), f2d, C.byref(nz))
    if rc == _lib.PRAD_E_INDEX:
        raise IndexError("Calculation of GLSZM Failed.")   # _cmatrices.c:372
    if rc < 0:
        _lib.raise_for(rc, "GLSZM")
    maxRegion = max(int(rc), 1)   # _cmatrices.c:390
    out = np.empty((Nvox, Ng, maxRegion), dtype=np.float64)
    rc = lib.prad_fill_glszm(_vptr(out), Nvox, int(Ng), maxRegion)
    if rc == _lib.PRAD_INDEX_ERROR:
        raise IndexError("Error filling GLSZM.")
    _lib.raise_for(rc, "GLSZM")
    return out


def calculate_glszm_compact(image, mask, Ng, Ns, force2D, force2Ddimension):
    """Segment-mode GLSZM with the all-zero size columns left out (no reference analogue at this boundary; it is
    the matrix glszm.py:118-131 ends up with).  -> (P float64 [1, Ng, k], sizes int32 [k] ascending).
    Host arrays are uploaded; device tensors are used in place."""
    import torch
    engine = _engine()
    if not _on_device(image):
        img, msk, _ = _parse_arrays(image, mask)
        dev = torch.device("cuda", torch.cuda.current_device())
        image, mask = torch.from_numpy(img).to(dev), torch.from_numpy(msk).to(dev)
    P, sizes = engine.glszm_compact(image, mask, int(Ng), int(Ns), force2D, force2Ddimension)
    return P.cpu().numpy()[None], sizes


# ---- fused voxel-based features of the other four texture classes (prad_voxel_texture_features_dev) ---------
_ZONE_LIKE = {
    "glrlm": (3, ["ShortRunEmphasis", "LongRunEmphasis", "GrayLevelNonUniformity", "GrayLevelNonUniformityNormalized",
                  "RunLengthNonUniformity", "RunLengthNonUniformityNormalized", "RunPercentage", "GrayLevelVariance",
                  "RunVariance", "RunEntropy", "LowGrayLevelRunEmphasis", "HighGrayLevelRunEmphasis",
                  "ShortRunLowGrayLevelEmphasis", "ShortRunHighGrayLevelEmphasis", "LongRunLowGrayLevelEmphasis",
                  "LongRunHighGrayLevelEmphasis"]),
    "glszm": (4, ["SmallAreaEmphasis", "LargeAreaEmphasis", "GrayLevelNonUniformity", "GrayLevelNonUniformityNormalized",
                  "SizeZoneNonUniformity", "SizeZoneNonUniformityNormalized", "ZonePercentage", "GrayLevelVariance",
                  "ZoneVariance", "ZoneEntropy", "LowGrayLevelZoneEmphasis", "HighGrayLevelZoneEmphasis",
                  "SmallAreaLowGrayLevelEmphasis", "SmallAreaHighGrayLevelEmphasis", "LargeAreaLowGrayLevelEmphasis",
                  "LargeAreaHighGrayLevelEmphasis"]),
    # None = slot of the shared numbering that is a deprecated feature in this class (gldm.py:228,311)
    "gldm": (1, ["SmallDependenceEmphasis", "LargeDependenceEmphasis", "GrayLevelNonUniformity", None,
                 "DependenceNonUniformity", "DependenceNonUniformityNormalized", None, "GrayLevelVariance",
                 "DependenceVariance", "DependenceEntropy", "LowGrayLevelEmphasis", "HighGrayLevelEmphasis",
                 "SmallDependenceLowGrayLevelEmphasis", "SmallDependenceHighGrayLevelEmphasis",
                 "LargeDependenceLowGrayLevelEmphasis", "LargeDependenceHighGrayLevelEmphasis"]),
    "ngtdm": (2, ["Coarseness", "Contrast", "Busyness", "Complexity", "Strength"]),
}
VOXEL_GLRLM_FEATURES = _ZONE_LIKE["glrlm"][1]
VOXEL_GLSZM_FEATURES = _ZONE_LIKE["glszm"][1]
VOXEL_GLDM_FEATURES = [f for f in _ZONE_LIKE["gldm"][1] if f]
VOXEL_NGTDM_FEATURES = _ZONE_LIKE["ngtdm"][1]


def voxel_texture_features(cls, image, mask, distances, Ng, force2D, force2Ddimension, kernelRadius, voxels, features,
                           alpha=0):
    """-> {feature name: float64 [Nvox]} of feature class `cls` ("glrlm" | "glszm" | "gldm" | "ngtdm") for the kernels
    centred on `voxels` (int [Nd, Nvox]).  Raises NotImplementedError when the fused kernels do not cover the request
    (unknown / deprecated feature, Ng > 255, more than 32 angles, kernels above 512 voxels): the caller then builds
    the matrices with calculate_* and uses the numpy formulas."""
    engine = _engine()
    family, table = _ZONE_LIKE[cls]
    missing = [f for f in features if f not in table]
    if missing:
        raise NotImplementedError("not available in the fused voxel kernel: %s" % ", ".join(missing))
    ids = [table.index(f) for f in features]
    vox = voxels if _on_device(voxels) else _to_device(np.ascontiguousarray(np.asarray(voxels).astype(np.intc, copy=False)))
    try:
        out = engine.voxel_texture_features(family, _to_device(image, integer=True), _to_device(mask), Ng, vox, ids,
                                            kernelRadius, force2D, force2Ddimension,
                                            [int(d) for d in np.asarray(distances).ravel()], alpha)
    except NotImplementedError:
        raise
    out = out.cpu().numpy()
    return {f: out[i] for i, f in enumerate(features)}


# ---- segment-mode features evaluated on the device matrices (prad_glcm_features_dev / prad_zone_matrix_features_dev)
def _angle_mean(per_angle, empty):
    """np.nanmean over the angles the reference keeps (all-empty angles are deleted, e.g. glcm.py:186-198)"""
    kept = per_angle[~empty] if empty.any() else per_angle
    if kept.shape[0] == 0:
        return np.full(per_angle.shape[1], np.nan)
    if not np.isnan(kept).any():
        return kept.mean(0)          # (what nanmean computes without a NaN: sum over the angles / their number)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        return np.nanmean(kept, 0)


def segment_features(cls, image, mask, Ng, features, distances=(1,), force2D=False, force2Ddimension=0, alpha=0,
                     symmetrical=True, Ns=None):
    """-> {feature name: float} of feature class `cls` ("glcm" | "glrlm" | "glszm" | "gldm" | "ngtdm") in segment mode with the
    matrix AND the formulas on the device; image / mask are device tensors (discretised levels, ROI).
    Raises NotImplementedError for features outside the fused set (MCC, deprecated ones)."""
    return segment_features_enqueue(cls, image, mask, Ng, features, distances, force2D, force2Ddimension, alpha,
                                    symmetrical, Ns, deferred=False)()


def segment_image_enqueue(levels, mask, Ng, Ns, requests, force2D=False, force2Ddimension=0):
    """The enqueue half of segment_features_enqueue / firstorder_stats_enqueue for SEVERAL classes of one derived image in
    one library call (engine.image_enqueue).  requests: {class key: dict} with "features" (names) and, per class,
    "symmetrical" (glcm), "alpha" (gldm), "raw" + "shift" (firstorder).  Returns (token, {class key: finish}); finish() as
    returned by the per-class functions, to be called after segment_image_wait(token).  Classes whose request the fused
    kernels do not cover are left out of the returned dict (the caller queues them one by one)."""
    engine = _engine()
    bits = {"glcm": engine.IMG_GLCM, "glrlm": engine.IMG_GLRLM, "gldm": engine.IMG_GLDM, "ngtdm": engine.IMG_NGTDM,
            "glszm": engine.IMG_GLSZM, "firstorder": engine.IMG_FIRSTORDER}
    classes, take = 0, {}
    for cls, rq in requests.items():
        if cls == "firstorder":
            take[cls] = rq
            classes |= bits[cls]
            continue
        table = VOXEL_GLCM_FEATURES if cls == "glcm" else _ZONE_LIKE[cls][1]
        feats = [f for f in rq["features"] if not (cls == "glcm" and f == "MCC")]
        if cls not in bits or any(f not in table for f in feats):
            continue
        take[cls] = dict(rq, table=table, feats=feats, mcc=(cls == "glcm" and "MCC" in rq["features"]))
        classes |= bits[cls] | (engine.IMG_MCC if take[cls]["mcc"] else 0)
    if not classes:
        return None, {}
    fo = take.get("firstorder")
    tok = engine.image_enqueue(levels, mask, fo["raw"] if fo else None, int(Ng), int(Ns), classes,
                               symmetric=take.get("glcm", {}).get("symmetrical", True), alpha=int(take.get("gldm", {}).get("alpha", 0)),
                               force2D=force2D, force2Ddimension=force2Ddimension,
                               voxelArrayShift=float(fo["shift"]) if fo else 0.0)
    # (with the image launcher the result block and its layout arrive with segment_image_wait: everything below reads them
    #  from the token when it runs, i.e. after the wait; a class the device route declined -- layout -1 -- then says so)
    def flags(k, count):
        res, lay = tok["res"], tok["layout"]
        return res[lay[k]:lay[k] + (count + 1) // 2 + 1].view(np.int32)[:count]

    def named(rq, vals):
        return dict(zip(rq["feats"], vals[rq["idx"]].tolist()))

    def part(k, count):
        res, lay = tok["res"], tok["layout"]
        if lay[k] < 0:
            raise NotImplementedError("declined by the device route")
        return res[lay[k]:lay[k] + count]

    for rq in take.values():
        if "table" in rq:
            rq["idx"] = np.array([rq["table"].index(f) for f in rq["feats"]], dtype=np.intp)
    out = {}
    if "glcm" in take:
        rq = take["glcm"]

        def fin_glcm(rq=rq):
            res, lay = tok["res"], tok["layout"]
            Na = lay[11]
            vals = part(0, Na * 23).reshape(Na, 23)
            r = named(rq, _angle_mean(vals, flags(1, Na) != 0)) if rq["feats"] else {}
            if rq["mcc"] and lay[2] >= 0 and res[lay[2] + Na] == 0:
                import warnings
                with np.errstate(invalid="ignore"), warnings.catch_warnings():
                    warnings.simplefilter("ignore", RuntimeWarning)
                    r["MCC"] = float(np.nanmean(res[lay[2]:lay[2] + Na]))
            return r
        out["glcm"] = fin_glcm
    if "glrlm" in take:
        def fin_glrlm(rq=take["glrlm"]):
            Na = tok["layout"][11]
            return named(rq, _angle_mean(part(3, Na * 16).reshape(Na, 16), flags(4, Na) != 0))
        out["glrlm"] = fin_glrlm
    if "gldm" in take:
        out["gldm"] = lambda rq=take["gldm"]: named(rq, _angle_mean(part(5, 16).reshape(1, 16), flags(6, 1) != 0))
    if "ngtdm" in take:
        out["ngtdm"] = lambda rq=take["ngtdm"]: named(rq, part(7, 5))
    if "glszm" in take:
        def fin_glszm(rq=take["glszm"]):
            vals = part(8, 17) if tok["layout"][8] >= 0 else None
            if vals is None or vals[16] != 0:       # the one-queue route or its device-side ranking declined / the reference's IndexError: the exact route
                P, sizes = engine.glszm_compact(levels, mask, int(Ng), Ns, force2D, force2Ddimension)
                if len(sizes) == 0:
                    raise NotImplementedError("no zones")
                return named(rq, _angle_mean(*engine.zone_matrix_features(P, sizes)))
            if flags(9, 1)[0] != 0:
                raise NotImplementedError("no zones")
            return named(rq, vals[:16])
        out["glszm"] = fin_glszm
    if fo is not None:
        def fin_fo():
            lay = tok["layout"]
            vals = part(10, 16) if lay[10] >= 0 else None
            if vals is None or vals[15] != 0:
                return engine.firstorder_stats(_to_device(fo["raw"]), _to_device(mask), fo["shift"])
            return dict(zip(engine.FIRSTORDER_FIELDS, vals[:15].tolist()))
        out["firstorder"] = fin_fo
    lay = tok.get("layout")
    if lay is not None:
        # issued by this thread (no launcher): what the device route declined is known already -- those classes are left out
        # and the caller queues them on its own; with the launcher the same is found out when the closure runs
        for cls, k in (("glcm", 0), ("glrlm", 3), ("gldm", 5), ("ngtdm", 7), ("glszm", 8), ("firstorder", 10)):
            if lay[k] < 0:
                out.pop(cls, None)
    return tok, out


def segment_image_wait(token):
    engine = _engine()
    return engine.image_wait(token)


SEGMENT_QUEUES = {"glcm": 0, "glrlm": 0, "gldm": 0, "ngtdm": 0, "glszm": 1, "firstorder": 2}


def segment_sync():
    """waits for everything segment_features_enqueue queued (inside segment_queue()); False when a queued call met a level
    outside [1, Ng] under the mask (the values are void: compute synchronously, which raises what the reference raises)"""
    engine = _engine()
    from . import _lib
    ok = True
    for k in sorted(set(SEGMENT_QUEUES.values())):
        try:
            with engine.side_queue(k, wait=False):
                engine.deferred_status()
        except _lib.DeferredLevelsError:
            ok = False
    return ok


def segment_mark(classes=None):
    """token behind everything queued inside segment_queue() so far (engine.deferred_mark on every side stream the
    classes use)"""
    engine = _engine()
    used = sorted(set(SEGMENT_QUEUES.get(c, 0) for c in (classes or SEGMENT_QUEUES)))
    marks = []
    for k in used:
        with engine.side_queue(k, wait=False):
            marks.append((k, engine.deferred_mark()))
    return marks


def segment_wait(token):
    """waits for the work in front of the token; False when its values are void (see segment_sync)"""
    engine = _engine()
    ok = True
    for k, mark in token:
        with engine.side_queue(k, wait=False):
            ok = engine.deferred_wait(mark) and ok
    return ok


def segment_queue(cls=None):
    """context manager around the enqueue calls of one feature class of a derived image: they go to the class's side stream
    (engine.side_queue) so that the classes evaluated synchronously meanwhile do not wait for them"""
    engine = _engine()
    return engine.side_queue(SEGMENT_QUEUES.get(cls, 0))


ENQUEUE_CLASSES = ("glcm", "glrlm", "gldm", "ngtdm", "glszm")


def segment_features_enqueue(cls, image, mask, Ng, features, distances=(1,), force2D=False, force2Ddimension=0, alpha=0,
                             symmetrical=True, Ns=None, deferred=True):
    """segment_features in two halves, for the case pipeline (featureextractor.computeFeatures): with deferred=True the
    matrix and feature kernels of `cls` are only ENQUEUED on the current stream, their values land in the library's
    result arena; the returned finish() -> {feature name: float} may be called once engine.deferred_status() has
    synchronised the stream (and not raised: a level outside [1, Ng] voids the values).  No reference analogue (the reference evaluates class after class on the host, base.py:181-198)."""
    engine = _engine()
    dist = [int(d) for d in np.asarray(distances).ravel()]
    if cls == "glcm":
        table = VOXEL_GLCM_FEATURES
    else:
        table = _ZONE_LIKE[cls][1]
    want_mcc = cls == "glcm" and "MCC" in features
    features = [f for f in features if not (cls == "glcm" and f == "MCC")]
    missing = [f for f in features if f not in table]
    if missing:
        raise NotImplementedError("not available in the fused segment kernels: %s" % ", ".join(missing))
    dfr = bool(deferred) and cls in ENQUEUE_CLASSES

    def named(vals):
        return {f: float(vals[table.index(f)]) for f in features}

    def mean_of(pair):
        return _angle_mean(pair[0], np.asarray(pair[1]) != 0)

    if cls == "glcm":
        if dist == [1]:
            g = _dev_pairs_runs(image, mask, Ng, force2D, force2Ddimension, "glcm", deferred=dfr)["glcm_dev"]
        else:
            g, _ = engine.glcm(image, mask, int(Ng), dist, force2D, force2Ddimension)
        pair = engine.glcm_features(g, symmetrical, deferred=dfr) if features else None
        mcc = None
        if want_mcc:
            try:     # (absent from the result when more than 64 grey levels occur: the caller's host route takes it)
                mcc = engine.glcm_mcc(g, symmetrical, deferred=dfr)
            except NotImplementedError:
                mcc = None

        def finish():
            res = named(mean_of(pair)) if features else {}
            if mcc is not None and not (dfr and mcc[-1] != 0):
                import warnings
                with np.errstate(invalid="ignore"), warnings.catch_warnings():
                    warnings.simplefilter("ignore", RuntimeWarning)
                    res["MCC"] = float(np.nanmean(mcc[:-1] if dfr else mcc))
            return res
        return finish
    if cls == "glrlm":
        r = _dev_pairs_runs(image, mask, Ng, force2D, force2Ddimension, "glrlm", deferred=dfr)["glrlm_dev"]
        pair = engine.zone_matrix_features(r, np.arange(1, r.shape[1] + 1), deferred=dfr)
    elif cls == "gldm":
        P = _dev_gldm_ngtdm(image, mask, Ng, alpha, dist, force2D, force2Ddimension, dfr)[0]
        pair = engine.zone_matrix_features(P, np.arange(1, P.shape[1] + 1), deferred=dfr)
    elif cls == "glszm":
        def three_calls():
            P, sizes = engine.glszm_compact(image, mask, int(Ng), Ns, force2D, force2Ddimension)
            if len(sizes) == 0:
                raise NotImplementedError("no zones")
            return named(mean_of(engine.zone_matrix_features(P, sizes)))
        try:        # zones, ranked sizes, compact matrix and formulas in one queue (no host round trip in between)
            vals, none = engine.glszm_features(image, mask, int(Ng), Ns, force2D, force2Ddimension, deferred=dfr)
        except NotImplementedError:
            return three_calls

        def finish():
            if vals[16] != 0:           # the device-side ranking declined / the reference's IndexError: the exact route
                return three_calls()
            if none[0] != 0:
                raise NotImplementedError("no zones")
            return named(vals[:16])
        return finish
    elif cls == "ngtdm":
        vals = engine.ngtdm_features(_dev_gldm_ngtdm(image, mask, Ng, None, dist, force2D, force2Ddimension, dfr)[1],
                                     deferred=dfr)
        return lambda: named(vals)
    else:
        raise NotImplementedError(cls)
    return lambda: named(mean_of(pair))


# ---- first-order statistics (no native code in the reference: radiomics/firstorder.py is numpy; here the ROI /
# ---- the kernels are reduced on the device, see include/pyradiomics_amd.h) ----------------------------------
FIRSTORDER_FEATURES = ["Energy", "TotalEnergy", "Entropy", "Minimum", "10Percentile", "90Percentile", "Maximum", "Mean",
                       "Median", "InterquartileRange", "Range", "MeanAbsoluteDeviation",
                       "RobustMeanAbsoluteDeviation", "RootMeanSquared", "StandardDeviation", "Skewness", "Kurtosis",
                       "Variance", "Uniformity"]


def _to_device(a, integer=False):
    import torch
    if _on_device(a):
        return a
    a = np.ascontiguousarray(a)
    if a.dtype == np.bool_:
        a = a.view(np.uint8)
    elif integer:
        a = a.astype(np.intc, copy=False)
    elif a.dtype not in (np.float32, np.float64, np.int32, np.int16):
        a = a.astype(np.float64)
    return torch.from_numpy(a).to(torch.device("cuda", torch.cuda.current_device()))


def firstorder_stats(image, mask, voxelArrayShift=0.0):
    """-> {Np, Energy, Minimum, P10, P25, Median, P75, P90, Maximum, Mean, MAD, rMAD, m2, m3, m4} of image[mask]
    (numpy arrays are uploaded, device tensors used in place)"""
    engine = _engine()
    return engine.firstorder_stats(_to_device(image), _to_device(mask), voxelArrayShift)


def firstorder_stats_enqueue(image, mask, roi_count, voxelArrayShift=0.0):
    """firstorder_stats in two halves for the case pipeline: the passes are queued on the current stream (no host round
    trip between them, engine.firstorder_stats_queue); the returned finish() -> dict may be called once the stream has
    been waited for and falls back to firstorder_stats when the queued chain declined (verdict word).
    NotImplementedError: not an image for the queue (integer dtype, small ROI)."""
    engine = _engine()
    img, msk = _to_device(image), _to_device(mask)
    vals = engine.firstorder_stats_queue(img, msk, int(roi_count), voxelArrayShift, deferred=True)

    def finish():
        if vals[15] != 0:
            return engine.firstorder_stats(img, msk, voxelArrayShift)
        return dict(zip(engine.FIRSTORDER_FIELDS, (float(v) for v in vals[:15])))
    return finish


def voxel_firstorder(image, mask, levels, voxels, kernelRadius, bbsize, force2D, force2Ddimension, voxelArrayShift,
                     voxelVolume, features):
    """-> {feature name: float64 [Nvox]} for the kernels centred on `voxels` (int [Nd, Nvox])"""
    engine = _engine()
    ids = [FIRSTORDER_FEATURES.index(f) for f in features]
    vox = _to_device(np.ascontiguousarray(np.asarray(voxels).astype(np.intc, copy=False))) if not _on_device(voxels) else voxels
    out = engine.voxel_firstorder(_to_device(image), _to_device(mask), _to_device(levels, integer=True), vox, ids,
                                  kernelRadius, force2D, force2Ddimension, bbsize, voxelArrayShift, voxelVolume)
    out = out.cpu().numpy()
    return {f: out[i] for i, f in enumerate(features)}


def generate_angles(size, distances, bidirectional, force2D, force2Ddimension):
    """-> int32 [Na, Nd];  _cmatrices.c:882-924"""
    size = np.ascontiguousarray(np.asarray(size).astype(np.intc, copy=False))
    if size.ndim != 1:
        raise ValueError("Expected a 1D array for size")
    return _build_angles(size, distances, bidirectional, int(force2Ddimension) if force2D else -1).copy()


# ---- fused voxel-based GLCM features (no reference analogue: replaces calculate_glcm + numpy feature math in
# ---- voxel mode, see include/pyradiomics_amd.h) -------------------------------------------------------------
VOXEL_GLCM_FEATURES = ["Autocorrelation", "JointAverage", "ClusterProminence", "ClusterShade", "ClusterTendency",
                       "Contrast", "Correlation", "DifferenceAverage", "DifferenceEntropy", "DifferenceVariance",
                       "JointEnergy", "JointEntropy", "Imc1", "Imc2", "Idm", "Idmn", "Id", "Idn", "InverseVariance",
                       "MaximumProbability", "SumAverage", "SumEntropy", "SumSquares"]


def voxel_glcm_features(image, mask, distances, Ng, force2D, force2Ddimension, kernelRadius, voxels, features,
                        symmetrical=True):
    """-> {feature name: float64 [Nvox]} for the kernels centred on `voxels` (int [Nd, Nvox]).
    Raises NotImplementedError when the fused kernel does not cover the request (Ng > 64, MCC, ...): the caller
    then uses calculate_glcm and the numpy feature formulas."""
    want_mcc = "MCC" in features
    all_features, features = features, [f for f in features if f != "MCC"]
    unknown = [f for f in features if f not in VOXEL_GLCM_FEATURES]
    if unknown:
        raise NotImplementedError("not available in the fused voxel kernel: %s" % ", ".join(unknown))
    img, msk, size, vox, Nvox, f2d, angles = _common(image, mask, distances, False, force2D, force2Ddimension,
                                                     kernelRadius, voxels)
    if vox is None:
        raise RuntimeError("voxel_glcm_features needs a voxel list")
    Na, Nd = angles.shape
    mcc = None
    if want_mcc:      # its own kernel (one Jacobi eigenvalue iteration per kernel and angle, csrc/kernels_mcc.h)
        mcc = np.empty(Nvox, dtype=np.float64)
        rc = _lib.load().prad_voxel_glcm_mcc(_vptr(img), _vptr(msk), _iptr(size), Nd, _iptr(angles), Na, int(Ng), Nvox,
                                             _vptr(vox), int(kernelRadius), f2d, 1 if symmetrical else 0, _vptr(mcc))
        _lib.raise_for(rc, "voxel GLCM MCC")
        if not features:
            return {"MCC": mcc}
    ids = np.array([VOXEL_GLCM_FEATURES.index(f) for f in features], dtype=np.intc)
    out = np.empty((len(ids), Nvox), dtype=np.float64)
    empty = np.empty(Nvox, dtype=np.uint32)
    anyne = np.zeros(1, dtype=np.uint32)
    rc = _lib.load().prad_voxel_glcm_features(_vptr(img), _vptr(msk), _iptr(size), Nd, _iptr(angles), Na, int(Ng),
                                              Nvox, _vptr(vox), int(kernelRadius), f2d, 1 if symmetrical else 0,
                                              _iptr(ids), len(ids), _vptr(out), _vptr(empty), _vptr(anyne))
    _lib.raise_for(rc, "voxel GLCM features")
    res = {f: out[i] for i, f in enumerate(features)}
    if "JointAverage" in res:
        # glcm.py:292 takes a plain mean over the angles kept for the batch: NaN wherever a kernel lacks an angle
        # that some other kernel of the batch has
        res["JointAverage"] = np.where((empty & anyne[0]) != 0, np.nan, res["JointAverage"])
    if mcc is not None:
        res["MCC"] = mcc
    return res
