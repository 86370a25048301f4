"""The tabulated wavelet filters (pyradiomics_amd/filters.py, mirrored in oracle/filters_oracle.py).  PyWavelets is not
installed in this environment and the reference tree vendors no coefficient table, so the tables are pinned by what an
orthonormal wavelet filter must satisfy (Daubechies' conditions): any wrong digit breaks them at the size of the error.
coif1 is additionally pinned by the reference's notebook outputs (tests/test_notebook_pin.py)."""
import numpy as np
import pytest

from pyradiomics_amd.filters import WAVELETS, wavelet_filters

MOMENTS = {"haar": 1, "db1": 1, "db2": 2, "sym2": 2, "db3": 3, "sym3": 3, "db4": 4, "sym4": 4, "db5": 5, "coif1": 2, "coif2": 4}


@pytest.mark.parametrize("name", WAVELETS)
def test_filter_is_an_orthonormal_wavelet_of_its_order(name):
    lo, hi = wavelet_filters(name)
    F = len(lo)
    assert F % 2 == 0 and len(hi) == F
    assert abs(lo.sum() - np.sqrt(2.0)) < 2e-11 and abs((lo * lo).sum() - 1.0) < 2e-11
    for m in range(1, F // 2):                                   # double-shift orthogonality
        assert abs((lo[2 * m:] * lo[:F - 2 * m]).sum()) < 2e-11, (name, m)
    assert np.array_equal(hi, np.array([(-1) ** (k + 1) * lo[F - 1 - k] for k in range(F)]))     # PyWavelets' QMF convention
    assert abs(hi.sum()) < 2e-11 and abs((lo * hi).sum()) < 1e-15
    k = np.arange(F, dtype=np.float64)
    for p in range(MOMENTS[name]):                               # vanishing moments of the high-pass
        assert abs((hi * k ** p).sum()) < 1e-8 * max(1.0, (F ** p)), (name, p)


@pytest.mark.parametrize("name", WAVELETS)
def test_product_table_equals_restatement_and_swt_preserves_energy(name):
    from oracle import filters_oracle as fo
    lo, hi = wavelet_filters(name)
    olo, ohi = fo.wavelet_filters(name)
    assert np.array_equal(lo, olo) and np.array_equal(hi, ohi)
    x = np.random.default_rng(1).standard_normal((6, 10, 16))
    for ax in range(3):                                          # one undecimated level: |a|^2 + |d|^2 = 2 |x|^2 along every axis
        a, d = fo.swt_axis(x, lo, ax), fo.swt_axis(x, hi, ax)
        assert abs((a * a).sum() + (d * d).sum() - 2.0 * (x * x).sum()) < 1e-9 * (x * x).sum()


@pytest.mark.parametrize("name", WAVELETS)
def test_table_equals_pywavelets_when_it_is_installed(name):
    """the identities above cannot tell a filter from its time reverse or a db phase from a sym phase; where
    PyWavelets is importable the tables are compared with pywt.Wavelet(name).dec_lo / dec_hi digit for digit (skipped in
    this image: pywt is not installed and there is no network)"""
    pywt = pytest.importorskip("pywt")
    w = pywt.Wavelet(name)
    lo, hi = wavelet_filters(name)
    np.testing.assert_allclose(lo, np.asarray(w.dec_lo), rtol=0, atol=5e-16)
    np.testing.assert_allclose(hi, np.asarray(w.dec_hi), rtol=0, atol=5e-16)
