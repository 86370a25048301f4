"""ctypes binding of include/pyradiomics_amd.h.  There is no fallback: if the HIP library is missing the
import fails loudly, and if no GPU is visible every calculate_* call raises."""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRAD_LIB") or os.path.join(_HERE, "csrc", "libpyradiomics_amd.so")  # PRAD_LIB: ablation builds

PRAD_OK = 1
PRAD_INDEX_ERROR = 0
PRAD_E_ARG, PRAD_E_HIP, PRAD_E_NOMEM, PRAD_E_UNSUPPORTED, PRAD_E_INDEX = -1, -2, -3, -4, -5

_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

# every symbol include/pyradiomics_amd.h declares: name -> (restype, argtypes)
_COMMON = [_vp, _vp, _ip, C.c_int, _ip, C.c_int]          # image, mask, size, Nd, angles, Na
_VOX = [C.c_int, _vp, C.c_int, C.c_int]                   # Nvox, voxels, kernelRadius, force2Ddim
SYMBOLS = {
    "prad_version": (C.c_char_p, []),
    "prad_last_error": (C.c_char_p, []),
    "prad_last_path": (C.c_char_p, []),
    "prad_last_variant": (C.c_char_p, []),
    "prad_device_count": (C.c_int, []),
    "prad_set_device": (C.c_int, [C.c_int]),
    "prad_get_device": (C.c_int, []),
    "prad_workspace_bytes": (C.c_longlong, []),
    "prad_release_workspace": (C.c_int, []),
    "prad_last_device_ms": (C.c_double, []),
    "prad_last_kernel_ms": (C.c_double, [C.c_char_p]),
    "prad_timing_begin": (C.c_int, []),
    "prad_timing_begin_only": (C.c_int, [C.c_char_p]),
    "prad_timing_count": (C.c_int, [C.c_char_p]),
    "prad_timing_ms": (C.c_double, [C.c_char_p]),
    "prad_timing_calls": (C.c_int, []),
    "prad_timing_end": (C.c_int, []),
    "prad_set_deferred": (C.c_int, [C.c_int]),
    "prad_set_lanes": (C.c_int, [C.c_int]),
    "prad_set_deferred_mode": (C.c_int, [C.c_int]),
    "prad_deferred_join": (C.c_int, [C.c_void_p]),
    "prad_result_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "prad_set_workspace": (C.c_int, [C.c_int]),
    "prad_image_enqueue_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, _ip, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_double, C.POINTER(C.c_void_p), _ip, _ip, _vp]),
    "prad_image_wait": (C.c_int, [C.c_int]),
    "prad_image_submit": (C.c_int, [_vp, _vp, _vp, C.c_int, _ip, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_double, _vp, _ip]),
    "prad_image_submit_result": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), _ip]),
    "prad_image_submit_wait": (C.c_int, [C.c_int]),
    "prad_image_submit_release": (C.c_int, []),
    "prad_deferred_status": (C.c_int, [C.c_void_p]),
    "prad_deferred_mark": (C.c_int, [C.POINTER(C.c_int), C.c_void_p]),
    "prad_get_angle_count": (C.c_int, [_ip, _ip, C.c_int, C.c_int, C.c_int, C.c_int]),
    "prad_build_angles": (C.c_int, [_ip, _ip, C.c_int, C.c_int, C.c_int, C.c_int, _ip]),
    "prad_calculate_glcm": (C.c_int, _COMMON + [C.c_int] + _VOX + [_vp]),
    "prad_calculate_glcm_dev": (C.c_int, _COMMON + [C.c_int] + _VOX + [_vp, _vp]),
    "prad_calculate_glrlm": (C.c_int, _COMMON + [C.c_int, C.c_int] + _VOX + [_vp]),
    "prad_calculate_glrlm_dev": (C.c_int, _COMMON + [C.c_int, C.c_int] + _VOX + [_vp, _vp]),
    "prad_calculate_glcm_glrlm": (C.c_int, _COMMON + [C.c_int, C.c_int] + _VOX + [_vp, _vp]),
    "prad_calculate_glcm_glrlm_dev": (C.c_int, _COMMON + [C.c_int, C.c_int] + _VOX + [_vp, _vp, _vp]),
    "prad_calculate_gldm": (C.c_int, _COMMON + [C.c_int, C.c_int] + _VOX + [_vp]),
    "prad_calculate_gldm_dev": (C.c_int, _COMMON + [C.c_int, C.c_int] + _VOX + [_vp, _vp]),
    "prad_calculate_ngtdm": (C.c_int, _COMMON + [C.c_int] + _VOX + [_vp]),
    "prad_calculate_ngtdm_dev": (C.c_int, _COMMON + [C.c_int] + _VOX + [_vp, _vp]),
    "prad_neigh_accumulate_dev": (C.c_int, [C.c_int] + _COMMON + [C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "prad_neigh_finalize_dev": (C.c_int, [C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),
    "prad_calculate_glszm": (C.c_int, _COMMON + [C.c_int, C.c_int] + _VOX + [C.POINTER(C.c_longlong)]),
    "prad_calculate_glszm_dev": (C.c_int, _COMMON + [C.c_int, C.c_int] + _VOX + [C.POINTER(C.c_longlong), _vp]),
    "prad_fill_glszm": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "prad_fill_glszm_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    "prad_glszm_zones": (C.c_longlong, [C.c_int, _ip, C.c_longlong]),
    "prad_glszm_sizes": (C.c_int, [_ip, C.c_int]),
    "prad_fill_glszm_compact_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "prad_voxel_glcm_features": (C.c_int, _COMMON + [C.c_int] + _VOX + [C.c_int, _ip, C.c_int, _vp, _vp, _vp]),
    "prad_voxel_glcm_features_dev": (C.c_int, _COMMON + [C.c_int] + _VOX + [C.c_int, _ip, C.c_int, _vp, _vp, _vp, _vp]),
    "prad_voxel_glcm_mcc": (C.c_int, _COMMON + [C.c_int] + _VOX + [C.c_int, _vp]),
    "prad_voxel_glcm_mcc_dev": (C.c_int, _COMMON + [C.c_int] + _VOX + [C.c_int, _vp, _vp]),
    "prad_glcm_mcc_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), _vp]),
    "prad_roi_minmax_dev": (C.c_int, [_vp, C.c_int, _vp, C.c_longlong, C.POINTER(C.c_double), _vp]),
    "prad_digitize_dev": (C.c_int, [_vp, C.c_int, _vp, C.c_longlong, C.POINTER(C.c_double), C.c_int, _vp, _ip, _vp]),
    "prad_digitize_counts_dev": (C.c_int, [_vp, C.c_int, _vp, C.c_longlong, C.POINTER(C.c_double), C.c_int, _vp, _ip,
                                          C.POINTER(C.c_longlong), _vp]),
    "prad_bincount_dev": (C.c_int, [_vp, C.c_int, _vp, C.c_longlong, C.c_int, _vp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(C.c_int), C.POINTER(C.c_longlong), _vp]),
    "prad_bincount_enqueue_dev": (C.c_int, [_vp, C.c_int, _vp, C.c_longlong, C.c_int, _vp, _ip, _vp]),
    "prad_bincount_wait": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                     C.POINTER(C.c_longlong)]),
    "prad_voxel_texture_features_dev": (C.c_int, [C.c_int, _vp, _vp, _ip, C.c_int, _ip, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  _vp, C.c_int, C.c_int, _ip, C.c_int, _vp, _vp]),
    "prad_glcm_features_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), _ip, _vp]),
    "prad_zone_matrix_features_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong,
                                                C.POINTER(C.c_double), C.POINTER(C.c_double), _ip, _vp]),
    "prad_ngtdm_features_dev": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), _vp]),
    "prad_calculate_gldm_ngtdm_dev": (C.c_int, [_vp, _vp, _ip, C.c_int, _ip, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "prad_glszm_features_dev": (C.c_int, [_vp, _vp, _ip, C.c_int, _ip, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), _ip,
                                          _vp]),
    "prad_resample_dev": (C.c_int, [_vp, C.c_int, _ip, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), _ip, C.c_int,
                                    _vp, _vp]),
    "prad_firstorder_dev": (C.c_int, [_vp, C.c_int, _vp, C.c_longlong, C.c_double, C.POINTER(C.c_double), _vp]),
    "prad_firstorder_queue_dev": (C.c_int, [_vp, C.c_int, _vp, C.c_longlong, C.c_longlong, C.c_double, C.POINTER(C.c_double),
                                            _vp]),
    "prad_voxel_firstorder_dev": (C.c_int, [_vp, C.c_int, _vp, _vp, _ip, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _ip,
                                            C.c_double, C.c_double, _ip, C.c_int, _vp, _vp]),
    "prad_level_counts_dev": (C.c_int, [_vp, _vp, C.c_longlong, C.c_int, C.POINTER(C.c_longlong), _vp]),
    "prad_swt_level1": (C.c_int, [_vp, _ip, C.c_int, _vp, _vp, C.c_int, _ip, C.c_int, _vp]),
    "prad_swt_level1_dev": (C.c_int, [_vp, _ip, C.c_int, _vp, _vp, C.c_int, _ip, C.c_int, _vp, _vp]),
    "prad_swt_level1_any_dev": (C.c_int, [_vp, C.c_int, _ip, C.c_int, _vp, _vp, C.c_int, _ip, C.c_int, _vp, _vp]),
    "prad_log": (C.c_int, [_vp, _ip, C.c_int, _vp, C.c_double, C.c_int, _vp]),
    "prad_log_dev": (C.c_int, [_vp, _ip, C.c_int, _vp, C.c_double, C.c_int, _vp, _vp]),
    "prad_log_multi_dev": (C.c_int, [_vp, _ip, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "prad_log_f64": (C.c_int, [_vp, _ip, C.c_int, _vp, C.c_double, C.c_int, _vp]),
    "prad_log_dev_f64": (C.c_int, [_vp, _ip, C.c_int, _vp, C.c_double, C.c_int, _vp, _vp]),
    "prad_log_multi_dev_f64": (C.c_int, [_vp, _ip, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
}

_lib = None


def load():
    """Returns the loaded CDLL (cached).  Raises ImportError when the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pyradiomics_amd: %s is missing. Build it with `python -m pyradiomics_amd._build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # torch ships its own libamdhip64 (same SONAME as /opt/rocm's).  Whichever is loaded first serves the whole
    # process, and two copies cannot both own the GPU, so when torch is installed it is imported BEFORE the
    # engine so that torch tensors and the engine share one HIP runtime (set PRAD_NO_TORCH=1 to skip).
    if "torch" not in sys.modules and not os.environ.get("PRAD_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:  # torch absent: the engine runs on /opt/rocm's runtime alone
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().prad_last_error().decode("utf-8", "replace")


def last_variant() -> str:
    return load().prad_last_variant().decode()


def last_path() -> str:
    return load().prad_last_path().decode()


PRAD_E_DEFERRED = -6


class DeferredLevelsError(RuntimeError):
    """prad_deferred_status: a deferred call met a level outside [1, Ng] under the mask (its outputs are void)"""


def raise_for(rc: int, what: str) -> None:
    """Maps a C status to the exception the reference wrapper raises (_cmatrices.c:219,566,714,864...)."""
    if rc == PRAD_OK:
        return
    if rc == PRAD_INDEX_ERROR:
        raise IndexError("Calculation of %s Failed." % what)
    msg = "%s: %s" % (what, last_error())
    if rc == PRAD_E_NOMEM:
        raise MemoryError(msg)
    if rc == PRAD_E_ARG:
        raise ValueError(msg)
    if rc == PRAD_E_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == PRAD_E_DEFERRED:
        raise DeferredLevelsError(msg)
    raise RuntimeError(msg)
