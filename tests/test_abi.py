"""The C-ABI library builds, loads and exports every symbol include/pyradiomics_amd.h declares; the host-only
entry points (angle enumeration) behave like the reference.  No compute calls here (no GPU in this tier)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "pyradiomics_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(prad_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from pyradiomics_amd import _build, _lib
    _build.build()          # no-op when the in-tree .so is current
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from pyradiomics_amd import _lib
    names = _header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "library does not export " + n
    assert sorted(_lib.SYMBOLS) == names, "ctypes table and header disagree"
    assert lib.prad_version().decode().startswith("pyradiomics_amd")


def test_no_cpu_fallback_without_device(lib):
    """without a HIP device every calculate_* call must fail loudly (never compute on the host)"""
    if lib.prad_device_count() > 0:
        pytest.skip("a GPU is visible")
    from pyradiomics_amd import cmatrices
    with pytest.raises(RuntimeError, match="no HIP device"):
        cmatrices.calculate_glcm(np.ones((3, 3, 3), int), np.ones((3, 3, 3), bool), [1], 1, False, 0)


def test_angles_match_oracle(lib, oracle_port):
    from pyradiomics_amd import cmatrices
    for bi in (0, 1):
        for size in [(5, 5, 5), (1, 5, 5), (2, 2, 2), (3, 1, 3), (4, 4), (9,), (2, 3, 2, 3), (512, 512, 512)]:
            for dist in ([1], [2], [1, 2], [3], [0]):
                for f2 in ((False, 0), (True, 0), (True, len(size) - 1)):
                    try:
                        want = oracle_port.generate_angles(size, dist, bi, *f2)
                    except RuntimeError:
                        with pytest.raises(RuntimeError):
                            cmatrices.generate_angles(size, dist, bi, *f2)
                        continue
                    got = cmatrices.generate_angles(size, dist, bi, *f2)
                    assert got.dtype == np.intc and np.array_equal(got, want)


def test_argument_validation_is_host_side(lib):
    """shape / rank / voxel-list errors are raised before any device work (same types as _cmatrices.c)"""
    from pyradiomics_amd import cmatrices as cm
    img = np.ones((4, 4, 4), int)
    with pytest.raises(ValueError):
        cm.calculate_glcm(img, np.ones((4, 4), bool), [1], 1, False, 0)
    with pytest.raises(ValueError):
        cm.calculate_glcm(img, np.ones((4, 4, 3), bool), [1], 1, False, 0)
    with pytest.raises(RuntimeError):
        cm.calculate_glcm(img, np.ones((4, 4, 4), bool), [1], 1, False, 0, 0, np.zeros((3, 1), int))
    with pytest.raises(RuntimeError):
        cm.calculate_glcm(img, np.ones((4, 4, 4), bool), [0], 1, False, 0)
    with pytest.raises(ValueError):
        cm.generate_angles(np.ones((2, 2)), [1], 0, 0, 0)
