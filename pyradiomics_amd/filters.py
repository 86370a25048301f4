"""Wavelet and Laplacian-of-Gaussian image types: the generator protocol of the reference's
radiomics/imageoperations.py (`get<Type>Image(inputImage, inputMask, **kwargs)` yielding
(image, imageTypeName, kwargs); featureextractor.py:371-379) with the filtering done on the MI355X through
prad_swt_level1 / prad_log (include/pyradiomics_amd.h).

Names follow the reference exactly: "wavelet-LLH" ... "wavelet-LLL" / "wavelet<k>-XYZ" (imageoperations.py:882-893)
with the first letter belonging to the x axis, and "log-sigma-<sigma with '.'->'-'>-mm-3D" (:827).
The third-party arithmetic behind both filters (PyWavelets' swtn, ITK's recursive Gaussian) is restated from its
published algorithms and pinned by the outputs the reference recorded in notebooks/helloFeatureClass.ipynb
(tests/test_notebook_pin.py: 198 brain1 values, wavelet <= 1e-9, LoG <= 1e-6 relative)."""
from __future__ import annotations

import ctypes as C
import logging

import numpy as np

from . import _lib
from .image import Image, as_array, as_image

logger = logging.getLogger(__name__)

# decomposition low-pass filters as tabulated by PyWavelets; dec_hi[k] = (-1)^(k+1) dec_lo[F-1-k]
_DEC_LO = {
    "coif1": [-0.01565572813546454, -0.0727326195128539, 0.38486484686420286, 0.8525720202122554,
              0.3378976624578092, -0.0727326195128539],
    "haar": [0.7071067811865476, 0.7071067811865476],
    "db1": [0.7071067811865476, 0.7071067811865476],
    "db2": [-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025],
    "sym2": [-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025],
    # round 5: more of PyWavelets' orthogonal families (the reference takes any pywt.Wavelet name, imageoperations.py:921).
    # PyWavelets is not installed here: the digits are as recalled from its tables and CHECKED by the identities an
    # orthonormal wavelet of that order must satisfy -- sum h = sqrt 2, sum h^2 = 1, double-shift orthogonality and the
    # vanishing moments of the high-pass, all to <= 1e-11, the accuracy pywt's own tables have (tests/test_wavelet_tables.py)
    "db3": [0.035226291882100656, -0.08544127388224149, -0.13501102001039084, 0.4598775021193313,
            0.8068915093133388, 0.3326705529509569],
    "sym3": [0.035226291882100656, -0.08544127388224149, -0.13501102001039084, 0.4598775021193313,
             0.8068915093133388, 0.3326705529509569],
    "db4": [-0.010597401784997278, 0.032883011666982945, 0.030841381835986965, -0.18703481171888114,
            -0.02798376941698385, 0.6308807679295904, 0.7148465705525415, 0.23037781330885523],
    "sym4": [-0.07576571478927333, -0.02963552764599851, 0.49761866763201545, 0.8037387518059161,
             0.29785779560527736, -0.09921954357684722, -0.012603967262037833, 0.0322231006040427],
    "db5": [0.003335725285001549, -0.012580751999015526, -0.006241490213011705, 0.07757149384006515,
            -0.03224486958502952, -0.24229488706619015, 0.13842814590110342, 0.7243085284385744,
            0.6038292697974729, 0.160102397974125],
    "coif2": [-0.0007205494453645122, -0.0018232088707029932, 0.0056114348193944995, 0.023680171946334084,
              -0.0594344186464569, -0.0764885990783064, 0.41700518442169254, 0.8127236354455423,
              0.3861100668211622, -0.06737255472196302, -0.04146493678175915, 0.016387336463522112],
}


WAVELETS = tuple(sorted(_DEC_LO))


def wavelet_filters(wavelet):
    """(dec_lo, dec_hi) float64 arrays for a wavelet name, or pass-through of a (dec_lo, dec_hi) pair"""
    if isinstance(wavelet, (tuple, list)) and len(wavelet) == 2:
        return np.asarray(wavelet[0], dtype=np.float64), np.asarray(wavelet[1], dtype=np.float64)
    if hasattr(wavelet, "dec_lo"):           # a pywt.Wavelet object
        return np.asarray(wavelet.dec_lo, dtype=np.float64), np.asarray(wavelet.dec_hi, dtype=np.float64)
    if wavelet not in _DEC_LO:
        raise NotImplementedError("wavelet %r is not tabulated here (known: %s); pass (dec_lo, dec_hi) instead"
                                  % (wavelet, ", ".join(sorted(_DEC_LO))))
    lo = np.array(_DEC_LO[wavelet], dtype=np.float64)
    F = len(lo)
    hi = np.array([(-1) ** (k + 1) * lo[F - 1 - k] for k in range(F)], dtype=np.float64)
    return lo, hi


def swtn_level1(data, lo, hi, axes):
    """pywt.swtn(data, wavelet, level=1, start_level=0, axes=axes)[0] on the device: dict key -> float64 array,
    keys in PyWavelets' order ('aaa', 'aad', ..., 'ddd')"""
    data = np.ascontiguousarray(data, dtype=np.float64)
    size = np.array(data.shape, dtype=np.intc)
    ax = np.array(axes, dtype=np.intc)
    out = np.empty((1 << len(ax),) + data.shape, dtype=np.float64)
    ip = C.POINTER(C.c_int)
    rc = _lib.load().prad_swt_level1(C.c_void_p(data.ctypes.data), size.ctypes.data_as(ip), data.ndim,
                                     C.c_void_p(lo.ctypes.data), C.c_void_p(hi.ctypes.data), len(lo),
                                     ax.ctypes.data_as(ip), len(ax), C.c_void_p(out.ctypes.data))
    _lib.raise_for(rc, "swt")
    keys = [""]
    for _ in ax:
        keys = [k + c for k in keys for c in "ad"]
    return {k: out[i] for i, k in enumerate(keys)}


def _swt3(array, axes, **kwargs):
    """imageoperations.py:899-970: pad odd dimensions by one wrapped sample, `level` single-level undecimated
    transforms (the un-dilated transform is re-applied to the previous approximation), crop the pad"""
    lo, hi = wavelet_filters(kwargs.get("wavelet", "coif1"))
    level = kwargs.get("level", 1)
    start_level = kwargs.get("start_level", 0)
    shape = array.shape
    data = np.pad(np.asarray(array).copy(), tuple((0, 1 if d % 2 else 0) for d in shape), "wrap")
    crop = tuple(slice(None, -1 if d % 2 else None) for d in shape)
    approx_key = "a" * len(axes)
    for _ in range(start_level):
        data = swtn_level1(data, lo, hi, axes)[approx_key].copy()
    ret = []
    for _ in range(start_level, start_level + level):
        dec = swtn_level1(data, lo, hi, axes)
        data = dec[approx_key].copy()
        ret.append({k.replace("a", "L").replace("d", "H"): v[crop].copy() for k, v in dec.items() if k != approx_key})
    return data[crop], ret


def _swt3_device(x, axes, **kwargs):
    """_swt3 with every intermediate in HBM: x and all results are torch tensors on the device"""
    import torch
    from . import engine
    lo, hi = wavelet_filters(kwargs.get("wavelet", "coif1"))
    level = kwargs.get("level", 1)
    start_level = kwargs.get("start_level", 0)
    shape = tuple(x.shape)
    from .engine import _DTYPE_CODES
    data = x if x.dtype in _DTYPE_CODES else x.to(torch.float64)    # (engine.swt_level1 widens to float64, in the fused kernel or as a copy)
    for d, n in enumerate(shape):                      # np.pad(..., 'wrap') by one sample on odd axes
        if n % 2:
            data = torch.cat([data, data.narrow(d, 0, 1)], dim=d)
    crop = tuple(slice(0, n) for n in shape)
    keys = [""]
    for _ in axes:
        keys = [k + c for k in keys for c in "ad"]
    approx_idx = keys.index("a" * len(axes))
    for _ in range(start_level):
        data = engine.swt_level1(data, lo, hi, axes)[approx_idx]
    ret = []
    for _ in range(start_level, start_level + level):
        sub = engine.swt_level1(data, lo, hi, axes)
        data = sub[approx_idx]
        ret.append({k.replace("a", "L").replace("d", "H"): sub[i][crop] for i, k in enumerate(keys) if i != approx_idx})
    return data[crop], ret


def getWaveletImage(inputImage, inputMask, **kwargs):
    """imageoperations.py:839-896.  By default the sub-bands stay in HBM (Images backed by device tensors; `.array`
    downloads on demand); `deviceResident=False` hands host arrays through prad_swt_level1 instead."""
    ref = as_image(inputImage)
    nd = len(ref.shape)
    axes = list(range(nd - 1, -1, -1))
    if kwargs.get("force2D", False):
        axes.remove(kwargs.get("force2Ddimension", 0))
    on_dev = kwargs.get("deviceResident", True)
    if on_dev:
        approx, ret = _swt3_device(ref.device_tensor(), tuple(axes), **kwargs)
    else:
        approx, ret = _swt3(ref.array, tuple(axes), **kwargs)

    def wrap(a):
        return ref.like(tensor=a) if on_dev else ref.like(a)
    for idx, wl in enumerate(ret, start=1):
        for name, dec in wl.items():
            yield wrap(dec), ("wavelet-%s" % name if idx == 1 else "wavelet%d-%s" % (idx, name)), kwargs
    tail = "L" * len(axes)
    yield wrap(approx), ("wavelet-%s" % tail if len(ret) == 1 else "wavelet%d-%s" % (len(ret), tail)), kwargs


def laplacian_recursive_gaussian(array, spacing_xyz, sigma, normalize=True):
    """sitk.LaplacianRecursiveGaussianImageFilter (NormalizeAcrossScale) on the device, host arrays in and out: float32,
    or float64 for a float64 input (the filter keeps the input's pixel type, imageoperations.py:824-830; its internal images
    are float32 either way, as in ITK)"""
    f64 = np.asarray(array).dtype == np.float64
    a = np.ascontiguousarray(array, dtype=np.float64 if f64 else np.float32)
    size = np.array(a.shape, dtype=np.intc)
    sp = np.array([float(s) for s in spacing_xyz][::-1], dtype=np.float64)
    out = np.empty(a.shape, dtype=a.dtype)
    fn = _lib.load().prad_log_f64 if f64 else _lib.load().prad_log
    rc = fn(C.c_void_p(a.ctypes.data), size.ctypes.data_as(C.POINTER(C.c_int)), a.ndim,
            C.c_void_p(sp.ctypes.data), float(sigma), 1 if normalize else 0, C.c_void_p(out.ctypes.data))
    _lib.raise_for(rc, "LoG")
    return out


def getLoGImage(inputImage, inputMask, **kwargs):
    """imageoperations.py:756-836"""
    ref = as_image(inputImage)
    on_dev = kwargs.get("deviceResident", True)
    size = np.array(ref.GetSize())
    spacing = np.array(ref.GetSpacing())
    if np.min(size) < 4:
        logger.warning("Image too small to apply LoG filter, size: %s", size)
        return
    ready = {}
    if on_dev:      # every admissible sigma in the same launches (engine.log_images), yielded in the reference's order below
        from . import engine
        ok = [s for s in kwargs.get("sigma", []) if s > 0.0 and np.all(size >= np.ceil(s / spacing) + 1)]
        ok = list(dict.fromkeys(ok))
        if ok:
            ready = dict(zip(ok, engine.log_images(ref.device_tensor(), spacing, ok, True)))
    for sigma in kwargs.get("sigma", []):
        if sigma > 0.0:
            if np.all(size >= np.ceil(sigma / spacing) + 1):
                name = "log-sigma-%s-mm-3D" % str(sigma).replace(".", "-")
                if on_dev:
                    yield ref.like(tensor=ready[sigma]), name, kwargs
                else:
                    yield ref.like(laplacian_recursive_gaussian(ref.array, spacing, sigma, True)), name, kwargs
            else:
                logger.warning("applyLoG: sigma(%s)/spacing(%s) + 1 must be greater than the size(%s) of the inputImage",
                               sigma, spacing, size)
        else:
            logger.warning("applyLoG: sigma must be greater than 0.0: %s", sigma)


# ---- intensity transforms (imageoperations.py:973-1073): explicit numpy formulas in the reference, evaluated here with
# ---- the same float64 operations on whichever side the image lives ------------------------------------------------
def _elementwise(inputImage, name, fn, **kwargs):
    ref = as_image(inputImage)
    if kwargs.get("deviceResident", True):
        import torch
        x = ref.device_tensor().to(torch.float64)
        return ref.like(tensor=fn(x, torch)), name, kwargs
    return ref.like(fn(ref.array.astype(np.float64), np)), name, kwargs


def getSquareImage(inputImage, inputMask, **kwargs):
    """f(x) = (c x)^2, c = 1 / sqrt(max |x|)  (imageoperations.py:973-994)"""
    def fn(x, xp):
        c = 1 / xp.sqrt(xp.abs(x).max())
        return (c * x) ** 2
    yield _elementwise(inputImage, "square", fn, **kwargs)


def getSquareRootImage(inputImage, inputMask, **kwargs):
    """f(x) = sign(x) sqrt(|x| c), c = max |x|  (imageoperations.py:997-1021)"""
    def fn(x, xp):
        c = xp.abs(x).max()
        return xp.sign(x) * xp.sqrt(xp.abs(x) * c)
    yield _elementwise(inputImage, "squareroot", fn, **kwargs)


def getLogarithmImage(inputImage, inputMask, **kwargs):
    """f(x) = sign(x) c log(|x| + 1), c = max |x| / max |log(|x| + 1)|  (imageoperations.py:1024-1049)"""
    def fn(x, xp):
        top = xp.abs(x).max()
        y = xp.sign(x) * xp.log(xp.abs(x) + 1)
        return y * (top / xp.abs(y).max())
    yield _elementwise(inputImage, "logarithm", fn, **kwargs)


def getExponentialImage(inputImage, inputMask, **kwargs):
    """f(x) = exp(c x), c = log(max |x|) / max |x|  (imageoperations.py:1052-1073)"""
    def fn(x, xp):
        top = xp.abs(x).max()
        return xp.exp((xp.log(top) / top) * x)
    yield _elementwise(inputImage, "exponential", fn, **kwargs)


def getGradientImage(inputImage, inputMask, **kwargs):
    """Gradient magnitude (imageoperations.py:1076-1091 on sitk.GradientMagnitudeImageFilter): central differences
    0.5 * (x[i+1] - x[i-1]) per axis with edge replication (ITK's ZeroFluxNeumann boundary), divided by the spacing
    unless `gradientUseSpacing` is false, root of the sum of squares in float64, float32 output.  ITK arithmetic
    restated from its documentation; no golden vector exists for this image type (parity unpinned)."""
    ref = as_image(inputImage)
    use_spacing = kwargs.get("gradientUseSpacing", True)
    spacing = ref.GetSpacing()[::-1]                       # numpy axis order
    if kwargs.get("deviceResident", True):
        import torch
        x = ref.device_tensor().to(torch.float64)
        acc = torch.zeros_like(x)
        for ax in range(x.dim()):
            n = x.shape[ax]
            if n == 1:
                continue
            idx = torch.arange(n, device=x.device)
            d = 0.5 * (x.index_select(ax, (idx + 1).clamp(max=n - 1)) - x.index_select(ax, (idx - 1).clamp(min=0)))
            if use_spacing:
                d = d / spacing[ax]
            acc += d * d
        yield ref.like(tensor=torch.sqrt(acc).to(torch.float32)), "gradient", kwargs
        return
    x = ref.array.astype(np.float64)
    acc = np.zeros_like(x)
    for ax in range(x.ndim):
        n = x.shape[ax]
        if n == 1:
            continue
        idx = np.arange(n)
        d = 0.5 * (np.take(x, np.minimum(idx + 1, n - 1), axis=ax) - np.take(x, np.maximum(idx - 1, 0), axis=ax))
        if use_spacing:
            d = d / spacing[ax]
        acc += d * d
    yield ref.like(np.sqrt(acc).astype(np.float32)), "gradient", kwargs


def getOriginalImage(inputImage, inputMask, **kwargs):
    """imageoperations.py:745-753"""
    yield inputImage, "original", kwargs
