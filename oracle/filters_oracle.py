"""CPU restatement of the third-party filter arithmetic behind the reference's wavelet / LoG image types.

TEST INFRASTRUCTURE ONLY (same rules as texture_oracle.c).

PARITY PINNED by outputs the reference itself recorded: notebooks/helloFeatureClass.ipynb (ipynb:1487-1559, :1609-1772)
stores 18 first-order values for each of LoG sigma 1 / 3 / 5 mm and the 8 coif1 sub-bands of brain1 as PyWavelets /
SimpleITK produced them; tests/golden/notebook_brain1.json (generator: tests/golden/make_notebook_golden.py) holds those
198 numbers and tests/test_notebook_pin.py checks this file against them (wavelet <= 1e-9 relative, LoG <= 1e-6 relative
or one float32 ulp of the image range for order statistics).
The arithmetic lives in wheels that are neither vendored in /root/reference nor installed here:
  * wavelet: PyWavelets >= 1.6.0 (pyproject.toml:38); call sites radiomics/imageoperations.py:921-935
             (`pywt.Wavelet`, `pywt.swtn(data, wavelet, level=1, start_level=0, axes=axes)`)
  * LoG:     SimpleITK >= 2.4.0 (pyproject.toml:37); call site imageoperations.py:824-830
             (`sitk.LaplacianRecursiveGaussianImageFilter`, NormalizeAcrossScale, sigma in mm)
and the reference's tests/ hold no usable golden vector for either (tests/test_wavelet.py compares the UNFILTERED image,
there is no LoG test) -- the notebook outputs above are the pin.  What follows restates the
published algorithms:

  swt (level 1, periodization)   out[o] = sum_k f[k] * x[(o + F/2 - k) mod N]          (PyWavelets
      `downsampling_convolution_periodization` with step 1, as used by `swt_axis` for level 1); sub-band keys are
      built axis by axis ('a' = dec_lo, 'd' = dec_hi) in the order of `axes`.
  recursive Gaussian             ITK `RecursiveGaussianImageFilter` / `RecursiveSeparableImageFilter`: 4th-order
      causal + anti-causal IIR with Deriche's coefficients as fitted by Farneback & Westin, zero / second order,
      edge-replicating boundary initialisation; Laplacian = sum over axes of (second order along the axis, zero
      order along the others) / spacing^2, times sigma^2 when normalising across scale; float32 images between
      the 1-D passes, float64 arithmetic inside a line (ITK's InternalRealType / RealType).
Sanity anchors available offline: perfect-reconstruction identities of the wavelet filters, and
scipy.ndimage.gaussian_laplace (an FIR approximation, agreement ~1e-2..1e-3, not a bit reference)."""
from __future__ import annotations

import math

import numpy as np

# decomposition filters as PyWavelets tabulates them (dec_lo; dec_hi follows the QMF rule below)
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


def wavelet_filters(name):
    lo = np.array(_DEC_LO[name], dtype=np.float64)
    F = len(lo)
    hi = np.array([(-1) ** (k + 1) * lo[F - 1 - k] for k in range(F)], dtype=np.float64)
    return lo, hi


def swt_axis(x, f, axis):
    """one undecimated, periodised analysis filter along `axis` (float64, sum over taps in ascending k)"""
    x = np.asarray(x, dtype=np.float64)
    N, F = x.shape[axis], len(f)
    out = np.zeros_like(x)
    idx = np.arange(N)
    for k in range(F):
        out += f[k] * np.take(x, (idx + F // 2 - k) % N, axis=axis)
    return out


def swtn_level1(data, lo, hi, axes):
    """dict key -> array, keys in PyWavelets' insertion order ('aaa', 'aad', ..., 'ddd')"""
    coeffs = {"": np.asarray(data, dtype=np.float64)}
    for ax in axes:
        nxt = {}
        for key, c in coeffs.items():
            nxt[key + "a"] = swt_axis(c, lo, ax)
            nxt[key + "d"] = swt_axis(c, hi, ax)
        coeffs = nxt
    return coeffs


def swt3(array, wavelet="coif1", level=1, start_level=0, axes=None):
    """restates radiomics/imageoperations.py:899-970 (_swt3): returns (approximation, [dict name->array per level])"""
    arr = np.asarray(array)
    if axes is None:
        axes = tuple(range(arr.ndim - 1, -1, -1))
    lo, hi = wavelet if isinstance(wavelet, tuple) else wavelet_filters(wavelet)
    shape = arr.shape
    data = np.pad(arr.copy(), tuple((0, 1 if d % 2 else 0) for d in shape), "wrap")
    crop = tuple(slice(None, -1 if d % 2 else None) for d in shape)
    for _ in range(start_level):
        data = swtn_level1(data, lo, hi, axes)["a" * len(axes)].copy()
    ret = []
    for _ in range(start_level, start_level + level):
        dec = swtn_level1(data, lo, hi, axes)
        data = dec["a" * len(axes)].copy()
        ret.append({k.replace("a", "L").replace("d", "H"): v[crop].copy() for k, v in dec.items()
                    if k != "a" * len(axes)})
    return data[crop], ret


# ---------------------------------------------------------------------------------------------------------
# ITK recursive Gaussian
# ---------------------------------------------------------------------------------------------------------
_W1, _L1, _W2, _L2 = 0.6681, -1.3932, 2.0787, -1.3732
_A1 = (1.3530, -0.6724, -1.3563)
_B1 = (1.8151, -3.4327, 5.2318)
_A2 = (-0.3531, 0.6724, 0.3446)
_B2 = (0.0902, 0.6100, -2.2355)


def _n_coefficients(sigmad, A1, B1, A2, B2):
    s1, s2 = math.sin(_W1 / sigmad), math.sin(_W2 / sigmad)
    c1, c2 = math.cos(_W1 / sigmad), math.cos(_W2 / sigmad)
    e1, e2 = math.exp(_L1 / sigmad), math.exp(_L2 / sigmad)
    N0 = A1 + A2
    N1 = e2 * (B2 * s2 - (A2 + 2 * A1) * c2)
    N1 += e1 * (B1 * s1 - (A1 + 2 * A2) * c1)
    N2 = (A1 + A2) * c2 * c1
    N2 -= B1 * c2 * s1 + B2 * c1 * s2
    N2 *= 2 * e1 * e2
    N2 += A2 * e1 * e1 + A1 * e2 * e2
    N3 = e2 * e1 * e1 * (B2 * s2 - A2 * c2)
    N3 += e1 * e2 * e2 * (B1 * s1 - A1 * c1)
    return (N0, N1, N2, N3), N0 + N1 + N2 + N3, N1 + 2 * N2 + 3 * N3, N1 + 4 * N2 + 9 * N3


def _d_coefficients(sigmad):
    c1, c2 = math.cos(_W1 / sigmad), math.cos(_W2 / sigmad)
    e1, e2 = math.exp(_L1 / sigmad), math.exp(_L2 / sigmad)
    D4 = e1 * e1 * e2 * e2
    D3 = -2 * c1 * e1 * e2 * e2
    D3 += -2 * c2 * e2 * e1 * e1
    D2 = 4 * c2 * c1 * e1 * e2
    D2 += e1 * e1 + e2 * e2
    D1 = -2 * (e2 * c2 + e1 * c1)
    SD = 1.0 + D1 + D2 + D3 + D4
    DD = D1 + 2 * D2 + 3 * D3 + 4 * D4
    ED = D1 + 4 * D2 + 9 * D3 + 16 * D4
    return (D1, D2, D3, D4), SD, DD, ED


def recursive_gaussian_coefficients(sigma, spacing, order, normalize_across_scale):
    """-> dict N[4], D[4], M[4], BN[4], BM[4] for a symmetric (order 0 or 2) filter"""
    sigmad = sigma / abs(spacing)
    D, SD, DD, ED = _d_coefficients(sigmad)
    if order == 0:
        N, SN, _, _ = _n_coefficients(sigmad, _A1[0], _B1[0], _A2[0], _B2[0])
        alpha0 = 2 * SN / SD - N[0]
        N = tuple(n / alpha0 for n in N)
    elif order == 2:
        scale = sigma * sigma if normalize_across_scale else 1.0
        N0s, SN0, DN0, EN0 = _n_coefficients(sigmad, _A1[0], _B1[0], _A2[0], _B2[0])
        N2s, SN2, DN2, EN2 = _n_coefficients(sigmad, _A1[2], _B1[2], _A2[2], _B2[2])
        beta = -(2 * SN2 - SD * N2s[0]) / (2 * SN0 - SD * N0s[0])
        N = tuple(n2 + beta * n0 for n2, n0 in zip(N2s, N0s))
        SN, DN, EN = SN2 + beta * SN0, DN2 + beta * DN0, EN2 + beta * EN0
        alpha2 = (EN * SD * SD - ED * SN * SD - 2 * DN * DD * SD + 2 * DD * DD * SN) / (SD * SD * SD)
        N = tuple(n * scale / alpha2 for n in N)
    else:
        raise ValueError("order must be 0 or 2")
    M = (N[1] - D[0] * N[0], N[2] - D[1] * N[0], N[3] - D[2] * N[0], -D[3] * N[0])
    SNn, SM = sum(N), sum(M)
    SDd = 1.0 + sum(D)
    BN = tuple(d * SNn / SDd for d in D)
    BM = tuple(d * SM / SDd for d in D)
    return {"N": N, "D": D, "M": M, "BN": BN, "BM": BM}


def _filter_lines(data, c):
    """ITK RecursiveSeparableImageFilter::FilterDataArray on the LAST axis of `data` (float64 in, float64 out)"""
    N0, N1, N2, N3 = c["N"]
    D1, D2, D3, D4 = c["D"]
    M1, M2, M3, M4 = c["M"]
    BN1, BN2, BN3, BN4 = c["BN"]
    BM1, BM2, BM3, BM4 = c["BM"]
    d = data
    ln = d.shape[-1]
    if ln < 4:
        raise ValueError("line too short for the recursive filter")
    s = np.empty_like(d)
    v1 = d[..., 0]
    s[..., 0] = v1 * N0 + v1 * N1 + v1 * N2 + v1 * N3
    s[..., 1] = d[..., 1] * N0 + v1 * N1 + v1 * N2 + v1 * N3
    s[..., 2] = d[..., 2] * N0 + d[..., 1] * N1 + v1 * N2 + v1 * N3
    s[..., 3] = d[..., 3] * N0 + d[..., 2] * N1 + d[..., 1] * N2 + v1 * N3
    s[..., 0] -= v1 * BN1 + v1 * BN2 + v1 * BN3 + v1 * BN4
    s[..., 1] -= s[..., 0] * D1 + v1 * BN2 + v1 * BN3 + v1 * BN4
    s[..., 2] -= s[..., 1] * D1 + s[..., 0] * D2 + v1 * BN3 + v1 * BN4
    s[..., 3] -= s[..., 2] * D1 + s[..., 1] * D2 + s[..., 0] * D3 + v1 * BN4
    for i in range(4, ln):
        s[..., i] = d[..., i] * N0 + d[..., i - 1] * N1 + d[..., i - 2] * N2 + d[..., i - 3] * N3
        s[..., i] -= s[..., i - 1] * D1 + s[..., i - 2] * D2 + s[..., i - 3] * D3 + s[..., i - 4] * D4
    out = s.copy()
    v2 = d[..., ln - 1]
    s[..., ln - 1] = v2 * M1 + v2 * M2 + v2 * M3 + v2 * M4
    s[..., ln - 2] = d[..., ln - 1] * M1 + v2 * M2 + v2 * M3 + v2 * M4
    s[..., ln - 3] = d[..., ln - 2] * M1 + d[..., ln - 1] * M2 + v2 * M3 + v2 * M4
    s[..., ln - 4] = d[..., ln - 3] * M1 + d[..., ln - 2] * M2 + d[..., ln - 1] * M3 + v2 * M4
    s[..., ln - 1] -= v2 * BM1 + v2 * BM2 + v2 * BM3 + v2 * BM4
    s[..., ln - 2] -= s[..., ln - 1] * D1 + v2 * BM2 + v2 * BM3 + v2 * BM4
    s[..., ln - 3] -= s[..., ln - 2] * D1 + s[..., ln - 1] * D2 + v2 * BM3 + v2 * BM4
    s[..., ln - 4] -= s[..., ln - 3] * D1 + s[..., ln - 2] * D2 + s[..., ln - 1] * D3 + v2 * BM4
    for i in range(ln - 4, 0, -1):
        s[..., i - 1] = d[..., i] * M1 + d[..., i + 1] * M2 + d[..., i + 2] * M3 + d[..., i + 3] * M4
        s[..., i - 1] -= s[..., i] * D1 + s[..., i + 1] * D2 + s[..., i + 2] * D3 + s[..., i + 3] * D4
    return out + s


def recursive_gaussian(image, axis, sigma, spacing, order, normalize_across_scale=False, real=np.float32):
    """one 1-D pass (itk::RecursiveGaussianImageFilter<TIn, Image<float>>): the line is read in the INPUT's own type and
    widened to float64 (RealType), float64 arithmetic inside the line, the result stored as `real` (float32: ITK's
    InternalRealType)"""
    c = recursive_gaussian_coefficients(sigma, spacing, order, normalize_across_scale)
    moved = np.moveaxis(np.asarray(image).astype(np.float64), axis, -1)
    return np.moveaxis(_filter_lines(moved, c), -1, axis).astype(real)


def laplacian_recursive_gaussian(array_zyx, spacing_xyz, sigma, normalize_across_scale=True):
    """ITK LaplacianRecursiveGaussianImageFilter on a numpy (z, y, x) array with SimpleITK (x, y, z) spacing.

    Pipeline of itkLaplacianRecursiveGaussianImageFilter.hxx: for every dimension `dim` (x, y, z) the DERIVATIVE filter
    (second order along dim, RecursiveGaussianImageFilter<InputImage, Image<float>>: it reads the input image itself) runs
    first, then the zero-order smoothing filters along the other dimensions in increasing ITK direction, float32 images
    between the passes (InternalRealType = float whatever the input type); the float32 cumulative image takes
    acc = float(acc + term / spacing[dim]^2); the result is cast to the input's real type at the end (float64 for float64
    inputs, float32 otherwise).
    PINNED (round 5): with exactly this order the Minimum, Maximum and Median of the sigma 1 / 3 / 5 mm images of brain1 --
    float32 values the reference's notebook recorded -- are reproduced BIT FOR BIT (tests/test_notebook_pin.py); with the
    smoothing passes in front of the derivative (the order of rounds 3-4) 8 of those 9 values are off by 1-4 float32 ulp."""
    arr = np.asarray(array_zyx)
    out_type = np.float64 if arr.dtype == np.float64 else np.float32
    real = np.float32
    nd = arr.ndim
    sp = [float(s) for s in spacing_xyz][::-1]   # per numpy axis
    acc = np.zeros(arr.shape, dtype=real)
    # ITK dimension order is x, y, z = numpy axes nd-1 ... 0
    for dim in range(nd - 1, -1, -1):
        cur = recursive_gaussian(arr, dim, sigma, sp[dim], 2, normalize_across_scale, real=real)
        for other in range(nd - 1, -1, -1):
            if other != dim:
                cur = recursive_gaussian(cur, other, sigma, sp[other], 0, real=real)
        acc = (acc.astype(np.float64) + cur.astype(np.float64) / (sp[dim] * sp[dim])).astype(real)
    return acc.astype(out_type)
