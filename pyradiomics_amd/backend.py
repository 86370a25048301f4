"""The `cMatrices` operator binding of this package (the analogue of radiomics/__init__.py:343-349).
Feature classes fetch the operator module through get(); by default that is the HIP engine
(pyradiomics_amd.cmatrices).  Tests may install a different object exposing the same six functions
(e.g. the CPU oracle) with set() to check the Python layer against the reference's golden matrices."""
from __future__ import annotations

_cmatrices = None


def get():
    global _cmatrices
    if _cmatrices is None:
        from . import cmatrices   # raises ImportError if the HIP library has not been built: no fallback
        _cmatrices = cmatrices
    return _cmatrices


def set(module) -> None:
    global _cmatrices
    _cmatrices = module
