"""Device-resident entry points: torch tensors that already live in HBM go straight to the `_dev` functions of
the C ABI (include/pyradiomics_amd.h); nothing is staged through the host.  torch is used only for device
memory and streams.  Used by bench.py, the batch driver and the voxel-map driver."""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np
import torch  # imported before the HIP library so both share one libamdhip64 instance

from . import _lib
from .cmatrices import _build_angles, _iptr


_tls = __import__("threading").local()


def _stream_ptr():
    p = getattr(_tls, "stream_ptr", None)      # (inside side_queue: torch.cuda.current_stream() costs ~15 us per call)
    return p if p is not None else C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _prep(image: torch.Tensor, mask: torch.Tensor):
    if not image.is_cuda or not mask.is_cuda:
        raise ValueError("engine.* expects CUDA/HIP tensors; use pyradiomics_amd.cmatrices for numpy input")
    if image.dtype != torch.int32:
        image = image.to(torch.int32)
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    elif mask.dtype != torch.uint8:
        mask = (mask != 0).view(torch.uint8)
    image = image.contiguous()
    mask = mask.contiguous()
    if image.shape != mask.shape:
        raise ValueError("Dimensions of image and mask do not match.")
    lib = _lib.load()
    dev = image.device.index if image.device.index is not None else torch.cuda.current_device()
    rc = lib.prad_set_device(dev)          # the library context is per thread: follow the tensor's device
    if rc != _lib.PRAD_OK:
        _lib.raise_for(rc, "set_device")
    size = np.array(image.shape, dtype=np.intc)
    return lib, image, mask, size


class _KeepList:
    """tensors the queued (deferred) calls of THIS host thread still use; released by deferred_status / deferred_wait"""

    def __init__(self):
        import threading
        self._tls = threading.local()

    def _l(self):
        l = getattr(self._tls, "l", None)
        if l is None:
            l = self._tls.l = []
        return l

    def append(self, x):
        self._l().append(x)

    def clear(self):
        self._l().clear()

    def __iter__(self):
        return iter(list(self._l()))

    def __len__(self):
        return len(self._l())


_NUMPY2 = int(np.__version__.split(".")[0]) >= 2
_deferred_keep = _KeepList()


def result_array(shape, dtype) -> np.ndarray:
    """numpy array over pinned memory of the library's result arena (prad_result_alloc): an enqueue-only feature call
    (deferred=True below) copies its values into it asynchronously; read it after deferred_status() / a stream sync, and
    before 4 MiB more have been allocated (include/pyradiomics_amd.h)"""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    p = C.c_void_p()
    _lib.raise_for(_lib.load().prad_result_alloc(max(n, 1) * dt.itemsize, C.byref(p)), "result arena")
    buf = (C.c_char * (max(n, 1) * dt.itemsize)).from_address(p.value)
    return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)


_side_streams: dict = {}


class side_queue:
    """`with side_queue(k):` -- calls inside go to side stream k (0..2) of the current device (one set per host thread) that
    first waits for everything queued on the current stream, under the library's workspace set k + 1 (prad_set_workspace:
    a second stream of one thread needs scratch buffers of its own).  The case pipeline queues GLCM / GLRLM / GLDM / NGTDM
    on stream 0, GLSZM on 1 and first order on 2 -- their many small kernels (13-workgroup formula kernels, single-workgroup
    glue kernels) overlap on the GPU -- while binning and the host-side classes keep the main stream.  wait=False: enter
    the same stream without the dependency (to mark or synchronise it)."""

    def __init__(self, which: int = 0, wait: bool = True):
        self.which, self.wait = int(which), wait

    def __enter__(self):
        import threading
        dev = torch.cuda.current_device()
        key = (threading.get_ident(), dev, self.which)
        s = _side_streams.get(key)
        if s is None:
            s = _side_streams[key] = torch.cuda.Stream(device=dev)
        if self.wait:
            s.wait_stream(torch.cuda.current_stream(dev))
        self._ctx = torch.cuda.stream(s)
        self._ctx.__enter__()
        _lib.raise_for(_lib.load().prad_set_workspace(self.which + 1), "workspace")
        _tls.stream_ptr = C.c_void_p(s.cuda_stream)
        return s

    def __exit__(self, *exc):
        _tls.stream_ptr = None
        _lib.load().prad_set_workspace(0)
        self._ctx.__exit__(*exc)
        return False


class _Deferred:
    """`with _Deferred(lib, on):` -- the library's deferred mode around one call"""

    def __init__(self, lib, on):
        self.lib, self.on = lib, on

    def __enter__(self):
        if self.on:
            _lib.raise_for(self.lib.prad_set_deferred(1), "deferred mode")

    def __exit__(self, *exc):
        if self.on:
            self.lib.prad_set_deferred(0)
        return False


def glcm_glrlm(image: torch.Tensor, mask: torch.Tensor, Ng: int, Nr: int | None = None, force2D: bool = False,
               force2Ddimension: int = 0, want_glcm: bool = True, want_glrlm: bool = True,
               out_glcm: torch.Tensor | None = None, out_glrlm: torch.Tensor | None = None, angles=None,
               deferred: bool = False):
    """GLCM [Ng,Ng,Na] and GLRLM [Ng,Nr,Na] (float64, on the device) of one discretised volume in segment
    mode, distance 1.  Returns (glcm, glrlm, angles).  `angles` (int32 [na, Nd]) restricts the sweep to a subset
    of the unidirectional distance-1 angles -- the angle shard of one rank when one segment is spread over
    several GPUs (batch.segment_matrices_sharded); the outputs then carry those na angles only.
    deferred=True only enqueues the kernels (no host synchronisation; consecutive volumes alternate between the
    library's two internal streams and share the GPU): the outputs are valid after deferred_status() -- which also
    reports whether a volume held masked levels outside [1, Ng] -- or, for work queued on the current stream, after
    deferred_join(); see include/pyradiomics_amd.h."""
    lib, image, mask, size = _prep(image, mask)
    f2d = int(force2Ddimension) if force2D else -1
    if angles is None:
        angles = _build_angles(size, None, False, f2d)
    else:
        angles = np.ascontiguousarray(np.asarray(angles, dtype=np.intc))
        if angles.ndim != 2 or angles.shape[1] != image.dim() or angles.shape[0] < 1:
            raise ValueError("angles must be int32 [na >= 1, Nd]")
    Na, Nd = angles.shape
    if Nr is None:
        Nr = int(max(image.shape))
    dev = image.device
    if want_glcm and out_glcm is None:
        out_glcm = torch.empty((Ng, Ng, Na), dtype=torch.float64, device=dev)
    if want_glrlm and out_glrlm is None:
        out_glrlm = torch.empty((Ng, Nr, Na), dtype=torch.float64, device=dev)
    if deferred:
        _lib.raise_for(lib.prad_set_deferred(1), "deferred mode")
        _deferred_keep.append((image, mask, out_glcm, out_glrlm))   # the lanes read/write these until deferred_status()
    try:
        rc = lib.prad_calculate_glcm_glrlm_dev(
            C.c_void_p(image.data_ptr()), C.c_void_p(mask.data_ptr()), _iptr(size), Nd, _iptr(angles), Na, int(Ng),
            int(Nr), 1, None, 0, f2d,
            C.c_void_p(out_glcm.data_ptr()) if want_glcm else None,
            C.c_void_p(out_glrlm.data_ptr()) if want_glrlm else None, _stream_ptr())
    finally:
        if deferred:
            lib.prad_set_deferred(0)
    _lib.raise_for(rc, "GLCM+GLRLM")
    return out_glcm, out_glrlm, angles


def deferred_status() -> None:
    """synchronises the current stream; raises if a deferred glcm_glrlm call since the last query saw levels outside
    [1, Ng] (its outputs are undefined: repeat that call with deferred=False)"""
    try:
        _lib.raise_for(_lib.load().prad_deferred_status(_stream_ptr()), "deferred GLCM+GLRLM")
    finally:
        _deferred_keep.clear()


IMG_GLCM, IMG_GLRLM, IMG_GLDM, IMG_NGTDM, IMG_GLSZM, IMG_FIRSTORDER, IMG_MCC = 1, 2, 4, 8, 16, 32, 64
_IMAGE_LAUNCHER = os.environ.get("PRAD_IMAGE_LAUNCHER", "1") != "0"     # (A/B switch: 0 = the calling thread issues the launches itself)


def image_enqueue(levels: torch.Tensor, mask: torch.Tensor, raw, Ng: int, Ns: int, classes: int, symmetric: bool = True,
                  alpha: int = 0, force2D: bool = False, force2Ddimension: int = 0, voxelArrayShift: float = 0.0):
    """every requested class (IMG_* bits) of ONE derived image queued by one library call (prad_image_enqueue_dev: sweeps,
    neighbourhood pass, GLSZM, first order and all formula kernels on the library's four side streams, two alternating sets of them; issued by the calling
    thread's launcher thread unless PRAD_IMAGE_LAUNCHER=0).  Returns a token
    {"res": float64 view of the result block, "layout": offsets (include/pyradiomics_amd.h), "ticket", "keep"}; the values
    are valid after image_wait(token)."""
    lib, levels, mask, size = _prep(levels, mask)
    code = 0
    if classes & IMG_FIRSTORDER:
        if raw.dtype not in _DTYPE_CODES:
            raw = raw.to(torch.float64)
        raw = raw.contiguous()
        code = _DTYPE_CODES[raw.dtype]
    if _IMAGE_LAUNCHER:
        # the ~65 launches of the image are issued by the calling thread's launcher thread (prad_image_submit): this thread goes
        # on with the next image's crop + binning and the previous image's values; result block and layout arrive with image_wait
        job = C.c_int(-1)
        rc = lib.prad_image_submit(C.c_void_p(levels.data_ptr()), C.c_void_p(mask.data_ptr()),
                                   C.c_void_p(raw.data_ptr()) if classes & IMG_FIRSTORDER else None, code, _iptr(size),
                                   levels.dim(), int(Ng), int(Ns), int(classes), 1 if symmetric else 0, int(alpha),
                                   int(force2Ddimension) if force2D else -1, float(voxelArrayShift), _stream_ptr(), C.byref(job))
        _lib.raise_for(rc, "image submit")
        return {"res": None, "layout": None, "job": int(job.value), "keep": (levels, mask, raw), "generation": _arena_gen.value}
    res_p = C.c_void_p()
    layout = (C.c_int * 16)()
    ticket = C.c_int(-1)
    rc = lib.prad_image_enqueue_dev(C.c_void_p(levels.data_ptr()), C.c_void_p(mask.data_ptr()),
                                    C.c_void_p(raw.data_ptr()) if classes & IMG_FIRSTORDER else None, code, _iptr(size),
                                    levels.dim(), int(Ng), int(Ns), int(classes), 1 if symmetric else 0, int(alpha),
                                    int(force2Ddimension) if force2D else -1, float(voxelArrayShift), C.byref(res_p), layout,
                                    C.byref(ticket), _stream_ptr())
    _lib.raise_for(rc, "image enqueue")
    n = max(int(layout[12]), 8)
    res = np.frombuffer((C.c_char * (8 * n)).from_address(res_p.value), dtype=np.float64, count=n)
    return {"res": res, "layout": [int(v) for v in layout], "ticket": int(ticket.value), "keep": (levels, mask, raw),
            "generation": _arena_gen.value}


def image_wait(token) -> bool:
    """waits for the work of an image_enqueue() token only; False when a queued call saw levels outside [1, Ng] (void)"""
    if "waited" in token:           # (a launcher token that was waited for already: its slot is free, the verdict is kept)
        return token["waited"]
    if token.get("job") is not None:
        lib = _lib.load()
        job, token["job"] = int(token["job"]), None
        token["waited"] = False
        res_p = C.c_void_p()
        layout = (C.c_int * 16)()
        rc0 = lib.prad_image_submit_result(job, C.byref(res_p), layout)
        rc = lib.prad_image_submit_wait(job)                  # (frees the slot, also after a failed submit: same code and message)
        token["keep"] = None
        if token.get("generation", _arena_gen.value) != _arena_gen.value:
            raise RuntimeError("release_workspace() was called while this image was queued: its result block is gone")
        _lib.raise_for(rc0, "image enqueue")
        n = max(int(layout[12]), 8)
        token["res"] = np.frombuffer((C.c_char * (8 * n)).from_address(res_p.value), dtype=np.float64, count=n)
        token["layout"] = [int(v) for v in layout]
        if rc == _lib.PRAD_E_DEFERRED:
            return False
        _lib.raise_for(rc, "image wait")
        token["waited"] = True
        return True
    if token.get("generation", _arena_gen.value) != _arena_gen.value:
        token["keep"] = None
        raise RuntimeError("release_workspace() was called while this image was queued: its result block is gone")
    rc = _lib.load().prad_image_wait(int(token["ticket"]))
    token["keep"] = None
    if rc == _lib.PRAD_E_DEFERRED:
        return False
    _lib.raise_for(rc, "image wait")
    return True


def deferred_mark():
    """token for deferred_wait(): everything queued on the current stream so far (a copy of the verdict word behind it, an
    event behind that); the tensors the queued calls use stay alive with the token"""
    flag = result_array((1,), np.intc)
    flag[0] = 0
    _lib.raise_for(_lib.load().prad_deferred_mark(_iptr(flag), _stream_ptr()), "deferred mark")
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    keep = list(_deferred_keep)
    _deferred_keep.clear()
    return flag, ev, keep


def deferred_wait(token) -> bool:
    """waits for the work in front of a deferred_mark() token only (later work on that stream keeps running); False when
    a deferred call in front of it saw levels outside [1, Ng] (the flag is then cleared through deferred_status)"""
    flag, ev, keep = token
    ev.synchronize()
    keep.clear()
    if flag[0] != 0:
        try:
            deferred_status()
        except _lib.DeferredLevelsError:
            pass
        return False
    return True


def deferred_join() -> None:
    """the current stream waits (on the device) for every deferred glcm_glrlm call issued so far: torch work queued
    afterwards may consume their outputs without a host synchronisation (deferred_status still reports the levels)"""
    _lib.raise_for(_lib.load().prad_deferred_join(_stream_ptr()), "deferred join")


def set_lanes(n: int) -> None:
    """internal streams the deferred whole-volume calls alternate between (0 = default 2, 1 = caller's stream only)"""
    _lib.raise_for(_lib.load().prad_set_lanes(int(n)), "lanes")


def set_deferred_mode(mode: int) -> None:
    """how deferred whole-volume calls overlap: 1 = two-stage pipeline on the caller's stream (volume N's pack rides in the
    walk launch of volume N-1; default), 0 = lanes (internal streams), -1 = environment default; flushes pending work"""
    _lib.raise_for(_lib.load().prad_set_deferred_mode(int(mode)), "deferred mode")


def timing_begin(only: str | None = None) -> None:
    """start keeping the HIP-event brackets of every following call of this thread; only="sweep": bracket that kernel family
    alone (two event records per call instead of ten -- every record costs the stream a few microseconds)"""
    if only:
        _lib.raise_for(_lib.load().prad_timing_begin_only(only.encode()), "timing")
    else:
        _lib.load().prad_timing_begin()


def timing_count(family: str | None = None) -> int:
    """brackets recorded for a kernel family since timing_begin (None: calls)"""
    return int(_lib.load().prad_timing_count(family.encode() if family else None))


def timing_ms(family: str | None = None) -> float:
    """device ms (HIP events on the launch stream) summed over every engine call since timing_begin()"""
    return float(_lib.load().prad_timing_ms(family.encode() if family else None))


def timing_calls() -> int:
    return int(_lib.load().prad_timing_calls())


def timing_end() -> None:
    _lib.load().prad_timing_end()


def pair_angles(shape, distances=(1,), force2D: bool = False, force2Ddimension: int = 0):
    """the unidirectional angle set GLCM (any distances) / GLRLM (distance 1) use for a volume of this shape"""
    size = np.asarray(shape, dtype=np.intc)
    return _build_angles(size, list(distances), False, int(force2Ddimension) if force2D else -1)


def glcm(image: torch.Tensor, mask: torch.Tensor, Ng: int, distances=(1,), force2D: bool = False,
         force2Ddimension: int = 0, angles=None):
    """GLCM [Ng, Ng, Na] float64 on the device for arbitrary distances (segment mode).  Returns (glcm, angles).
    `angles` restricts the call to a subset of the angle set (see glcm_glrlm)."""
    lib, image, mask, size = _prep(image, mask)
    f2d = int(force2Ddimension) if force2D else -1
    if angles is None:
        angles = _build_angles(size, list(distances), False, f2d)
    else:
        angles = np.ascontiguousarray(np.asarray(angles, dtype=np.intc))
        if angles.ndim != 2 or angles.shape[1] != image.dim() or angles.shape[0] < 1:
            raise ValueError("angles must be int32 [na >= 1, Nd]")
    Na, Nd = angles.shape
    out = torch.empty((Ng, Ng, Na), dtype=torch.float64, device=image.device)
    rc = lib.prad_calculate_glcm_dev(C.c_void_p(image.data_ptr()), C.c_void_p(mask.data_ptr()), _iptr(size), Nd,
                                     _iptr(angles), Na, int(Ng), 1, None, 0, f2d, C.c_void_p(out.data_ptr()),
                                     _stream_ptr())
    _lib.raise_for(rc, "GLCM")
    return out, angles


def glcm_features(glcm: torch.Tensor, symmetric: bool = True, deferred: bool = False):
    """the 23 sum-type GLCM features (order of cmatrices.VOXEL_GLCM_FEATURES) per angle from the raw device matrix
    [Ng, Ng, Na]: (float64 numpy [Na, 23], bool numpy [Na] = angle empty).
    deferred=True only enqueues (kernel + copies into the result arena): returns (values, int32 flags != 0 = empty), both
    valid after deferred_status() / a synchronisation of the current stream."""
    lib = _lib.load()
    glcm = glcm.contiguous()
    Ng, _, Na = glcm.shape
    lib.prad_set_device(glcm.device.index or 0)
    alloc = result_array if deferred else np.empty
    out = alloc((Na, 23), np.float64)
    empty = alloc((Na,), np.intc)
    with _Deferred(lib, deferred):
        rc = lib.prad_glcm_features_dev(C.c_void_p(glcm.data_ptr()), int(Ng), int(Na), 1 if symmetric else 0,
                                        out.ctypes.data_as(C.POINTER(C.c_double)), _iptr(empty), _stream_ptr())
    _lib.raise_for(rc, "GLCM features")
    if deferred:
        _deferred_keep.append(glcm)
        return out, empty
    return out, empty != 0


def glcm_mcc(glcm: torch.Tensor, symmetric: bool = True, deferred: bool = False):
    """per-angle MCC (glcm.py:665-707) from the raw device matrix [Ng, Ng, Na]: float64 numpy [Na], NaN for an angle
    without pairs.  Raises NotImplementedError when more than 64 grey levels occur (host route).
    deferred=True only enqueues: returns float64 [Na + 1] in the result arena, valid after deferred_status(); its last
    entry != 0 stands for that NotImplementedError (the values are void then)."""
    lib = _lib.load()
    glcm = glcm.contiguous()
    Ng, _, Na = glcm.shape
    lib.prad_set_device(glcm.device.index or 0)
    out = result_array((Na + 1,), np.float64) if deferred else np.empty(Na, dtype=np.float64)
    with _Deferred(lib, deferred):
        rc = lib.prad_glcm_mcc_dev(C.c_void_p(glcm.data_ptr()), int(Ng), int(Na), 1 if symmetric else 0,
                                   out.ctypes.data_as(C.POINTER(C.c_double)), _stream_ptr())
    _lib.raise_for(rc, "GLCM MCC")
    if deferred:
        _deferred_keep.append(glcm)
    return out


def voxel_glcm_mcc(image: torch.Tensor, mask: torch.Tensor, Ng: int, voxels: torch.Tensor, kernelRadius: int = 1,
                   force2D: bool = False, force2Ddimension: int = 0, symmetrical: bool = True, distances=(1,)):
    """voxel-based MCC map, everything device-resident: float64 tensor [Nvox]"""
    lib, image, mask, size = _prep(image, mask)
    f2d = int(force2Ddimension) if force2D else -1
    angles = _build_angles(size, list(distances), False, f2d)
    Na, Nd = angles.shape
    vox = voxels.to(torch.int32).contiguous()
    if vox.dim() != 2 or vox.shape[0] != Nd:
        raise RuntimeError("Expecting voxel indices array to be 2-dimensional")
    Nvox = int(vox.shape[1])
    out = torch.empty(Nvox, dtype=torch.float64, device=image.device)
    rc = lib.prad_voxel_glcm_mcc_dev(
        C.c_void_p(image.data_ptr()), C.c_void_p(mask.data_ptr()), _iptr(size), Nd, _iptr(angles), Na, int(Ng), Nvox,
        C.c_void_p(vox.data_ptr()), int(kernelRadius), f2d, 1 if symmetrical else 0, C.c_void_p(out.data_ptr()),
        _stream_ptr())
    _lib.raise_for(rc, "voxel GLCM MCC")
    return out


def zone_matrix_features(P: torch.Tensor, jvals, deferred: bool = False):
    """the 16 features GLRLM / GLSZM / GLDM share, per angle, from a device count matrix [Ni, Nj] or [Ni, Nj, Na] with
    level values 1..Ni and size values `jvals` [Nj]: (float64 numpy [Na, 16], bool numpy [Na] = matrix empty).
    deferred=True: enqueue only, see glcm_features."""
    lib = _lib.load()
    if P.dim() == 2:
        P = P.unsqueeze(2)
    Ni, Nj, Na = P.shape
    si, sj, sa = P.stride()
    lib.prad_set_device(P.device.index or 0)
    jv = np.ascontiguousarray(jvals, dtype=np.float64)
    if jv.shape != (Nj,):
        raise ValueError("jvals must have one entry per column")
    alloc = result_array if deferred else np.empty
    out = alloc((Na, 16), np.float64)
    empty = alloc((Na,), np.intc)
    with _Deferred(lib, deferred):
        rc = lib.prad_zone_matrix_features_dev(C.c_void_p(P.data_ptr()), int(Ni), int(Nj), int(Na), int(si), int(sj),
                                               int(sa), jv.ctypes.data_as(C.POINTER(C.c_double)),
                                               out.ctypes.data_as(C.POINTER(C.c_double)), _iptr(empty), _stream_ptr())
    _lib.raise_for(rc, "zone matrix features")
    if deferred:
        _deferred_keep.append(P)
        return out, empty
    return out, empty != 0


def ngtdm_features(P: torch.Tensor, deferred: bool = False) -> np.ndarray:
    """Coarseness, Contrast, Busyness, Complexity, Strength from the device NGTDM [Ng, 3] (deferred=True: enqueue only,
    see glcm_features)"""
    lib = _lib.load()
    P = P.contiguous()
    lib.prad_set_device(P.device.index or 0)
    out = result_array((5,), np.float64) if deferred else np.empty(5, dtype=np.float64)
    with _Deferred(lib, deferred):
        rc = lib.prad_ngtdm_features_dev(C.c_void_p(P.data_ptr()), int(P.shape[0]),
                                         out.ctypes.data_as(C.POINTER(C.c_double)), _stream_ptr())
    _lib.raise_for(rc, "NGTDM features")
    if deferred:
        _deferred_keep.append(P)
    return out


def resample(image: torch.Tensor, start, step, newsize, interpolator: int = 3) -> torch.Tensor:
    """ITK-style resampling on the device (prad_resample_dev): output voxel o along axis d reads the input at continuous
    index start[d] + o * step[d]; interpolator 0 nearest / 1 linear / 3 cubic B-spline; same dtype out"""
    lib = _lib.load()
    if image.dtype not in _DTYPE_CODES:
        image = image.to(torch.float64)
    image = image.contiguous()
    lib.prad_set_device(image.device.index or 0)
    nd = image.dim()
    size = np.array(image.shape, dtype=np.intc)
    st = np.ascontiguousarray(start, dtype=np.float64)
    sp = np.ascontiguousarray(step, dtype=np.float64)
    ns = np.ascontiguousarray(newsize, dtype=np.intc)
    out = torch.empty(tuple(int(v) for v in ns), dtype=image.dtype, device=image.device)
    dp = C.POINTER(C.c_double)
    rc = lib.prad_resample_dev(C.c_void_p(image.data_ptr()), _DTYPE_CODES[image.dtype], _iptr(size), nd,
                               st.ctypes.data_as(dp), sp.ctypes.data_as(dp), _iptr(ns), int(interpolator),
                               C.c_void_p(out.data_ptr()), _stream_ptr())
    _lib.raise_for(rc, "resample")
    return out


def workspace_bytes() -> int:
    """device bytes of scratch the library currently holds for this thread"""
    return int(_lib.load().prad_workspace_bytes())


class _Generation(threading.local):      # (the library's workspace is per host thread, so is its generation)
    value = 0


_arena_gen = _Generation()   # bumped by release_workspace(): result-arena views and image tickets issued before are void


def release_workspace() -> None:
    """free the library's cached scratch buffers (they are re-created on demand).  Whatever was queued and not yet collected
    -- image_enqueue tokens, result_array views -- is void afterwards: image_wait() on such a token raises"""
    _arena_gen.value += 1
    _lib.raise_for(_lib.load().prad_release_workspace(), "release_workspace")
    if _IMAGE_LAUNCHER:
        _lib.raise_for(_lib.load().prad_image_submit_release(), "release_workspace (image launcher)")


def last_device_ms() -> float:
    return float(_lib.load().prad_last_device_ms())


def last_kernel_ms(family: str | None = None) -> float:
    return float(_lib.load().prad_last_kernel_ms(family.encode() if family else None))


def last_path() -> str:
    return _lib.last_path()


def last_variant() -> str:
    """which sweep kernels served the last GLCM / GLRLM call: "fw", "fw2" or "lines" """
    return _lib.last_variant()


def _neigh_common(image, mask, distances, force2D, force2Ddimension):
    lib, image, mask, size = _prep(image, mask)
    f2d = int(force2Ddimension) if force2D else -1
    angles = _build_angles(size, distances, True, f2d)
    return lib, image, mask, size, f2d, angles


def gldm(image: torch.Tensor, mask: torch.Tensor, Ng: int, alpha: int = 0, distances=(1,), force2D: bool = False,
         force2Ddimension: int = 0, deferred: bool = False) -> torch.Tensor:
    """GLDM [Ng, 2*Na+1] float64 on the device (segment mode).  deferred=True only enqueues the kernels: levels outside
    [1, Ng] are reported by deferred_status() instead of an exception here."""
    lib, image, mask, size, f2d, angles = _neigh_common(image, mask, list(distances), force2D, force2Ddimension)
    Na, Nd = angles.shape
    out = torch.empty((Ng, 2 * Na + 1), dtype=torch.float64, device=image.device)
    with _Deferred(lib, deferred):
        rc = lib.prad_calculate_gldm_dev(C.c_void_p(image.data_ptr()), C.c_void_p(mask.data_ptr()), _iptr(size), Nd,
                                         _iptr(angles), Na, int(Ng), int(alpha), 1, None, 0, f2d,
                                         C.c_void_p(out.data_ptr()), _stream_ptr())
    _lib.raise_for(rc, "GLDM")
    if deferred:
        _deferred_keep.append((image, mask, out))
    return out


def ngtdm(image: torch.Tensor, mask: torch.Tensor, Ng: int, distances=(1,), force2D: bool = False,
          force2Ddimension: int = 0, deferred: bool = False) -> torch.Tensor:
    """NGTDM [Ng, 3] float64 on the device (segment mode); deferred=True: see gldm."""
    lib, image, mask, size, f2d, angles = _neigh_common(image, mask, list(distances), force2D, force2Ddimension)
    Na, Nd = angles.shape
    out = torch.empty((Ng, 3), dtype=torch.float64, device=image.device)
    with _Deferred(lib, deferred):
        rc = lib.prad_calculate_ngtdm_dev(C.c_void_p(image.data_ptr()), C.c_void_p(mask.data_ptr()), _iptr(size), Nd,
                                          _iptr(angles), Na, int(Ng), 1, None, 0, f2d, C.c_void_p(out.data_ptr()),
                                          _stream_ptr())
    _lib.raise_for(rc, "NGTDM")
    if deferred:
        _deferred_keep.append((image, mask, out))
    return out


def gldm_ngtdm(image: torch.Tensor, mask: torch.Tensor, Ng: int, alpha: int = 0, distances=(1,), force2D: bool = False,
               force2Ddimension: int = 0, deferred: bool = False):
    """(GLDM [Ng, 2*Na+1], NGTDM [Ng, 3]) of one segment from ONE pass over the neighbourhoods
    (prad_calculate_gldm_ngtdm_dev); deferred=True: see gldm"""
    lib, image, mask, size, f2d, angles = _neigh_common(image, mask, list(distances), force2D, force2Ddimension)
    Na, Nd = angles.shape
    g = torch.empty((Ng, 2 * Na + 1), dtype=torch.float64, device=image.device)
    n = torch.empty((Ng, 3), dtype=torch.float64, device=image.device)
    with _Deferred(lib, deferred):
        rc = lib.prad_calculate_gldm_ngtdm_dev(C.c_void_p(image.data_ptr()), C.c_void_p(mask.data_ptr()), _iptr(size), Nd,
                                               _iptr(angles), Na, int(Ng), int(alpha), C.c_void_p(g.data_ptr()),
                                               C.c_void_p(n.data_ptr()), _stream_ptr())
    _lib.raise_for(rc, "GLDM+NGTDM")
    if deferred:
        _deferred_keep.append((image, mask, g, n))
    return g, n


NEIGH_GLDM, NEIGH_NGTDM = 0, 1


def neigh_angles(shape, distances=(1,), force2D: bool = False, force2Ddimension: int = 0):
    """the bidirectional angle set GLDM / NGTDM use for a volume of this shape (int32 [Na, Nd])"""
    size = np.asarray(shape, dtype=np.intc)
    return _build_angles(size, list(distances), True, int(force2Ddimension) if force2D else -1)


def neigh_accumulate(family: int, image: torch.Tensor, mask: torch.Tensor, Ng: int, z_lo: int, z_hi: int,
                     alpha: int = 0, distances=(1,), force2D: bool = False, force2Ddimension: int = 0):
    """Integer accumulators int64 [Ng, Na+1] (on the device) of GLDM (family NEIGH_GLDM) or NGTDM (NEIGH_NGTDM) for
    the centre voxels in planes z_lo <= z < z_hi of a 3-D volume; neighbours come from the whole volume.  The
    accumulators of disjoint plane ranges sum to those of the whole volume (prad_neigh_accumulate_dev)."""
    lib, image, mask, size, f2d, angles = _neigh_common(image, mask, list(distances), force2D, force2Ddimension)
    Na, Nd = angles.shape
    acc = torch.empty((Ng, Na + 1), dtype=torch.int64, device=image.device)
    rc = lib.prad_neigh_accumulate_dev(int(family), C.c_void_p(image.data_ptr()), C.c_void_p(mask.data_ptr()),
                                       _iptr(size), Nd, _iptr(angles), Na, int(Ng), int(alpha), int(z_lo), int(z_hi),
                                       C.c_void_p(acc.data_ptr()), _stream_ptr())
    _lib.raise_for(rc, "GLDM" if family == NEIGH_GLDM else "NGTDM")
    return acc


def neigh_finalize(family: int, acc: torch.Tensor) -> torch.Tensor:
    """GLDM [Ng, 2*Na+1] / NGTDM [Ng, 3] float64 (on the device) from summed accumulators [Ng, Na+1]"""
    lib = _lib.load()
    acc = acc.contiguous()
    if acc.dtype != torch.int64 or acc.dim() != 2:
        raise ValueError("acc must be int64 [Ng, Na+1]")
    Ng, W = acc.shape
    Na = W - 1
    lib.prad_set_device(acc.device.index or 0)
    out = torch.empty((Ng, 2 * Na + 1) if family == NEIGH_GLDM else (Ng, 3), dtype=torch.float64, device=acc.device)
    rc = lib.prad_neigh_finalize_dev(int(family), C.c_void_p(acc.data_ptr()), int(Ng), int(Na),
                                     C.c_void_p(out.data_ptr()), _stream_ptr())
    _lib.raise_for(rc, "GLDM" if family == NEIGH_GLDM else "NGTDM")
    return out


def glszm(image: torch.Tensor, mask: torch.Tensor, Ng: int, Ns: int | None = None, force2D: bool = False,
          force2Ddimension: int = 0) -> torch.Tensor:
    """GLSZM [Ng, maxRegion] float64 on the device (segment mode); Ns defaults to the number of masked voxels."""
    lib, image, mask, size, f2d, angles = _neigh_common(image, mask, None, force2D, force2Ddimension)
    Na, Nd = angles.shape
    if Ns is None:
        Ns = int(mask.sum().item())
    nz = C.c_longlong(0)
    rc = lib.prad_calculate_glszm_dev(C.c_void_p(image.data_ptr()), C.c_void_p(mask.data_ptr()), _iptr(size), Nd,
                                      _iptr(angles), Na, int(Ng), int(Ns), 1, None, 0, f2d, C.byref(nz),
                                      _stream_ptr())
    if rc == _lib.PRAD_E_INDEX:
        raise IndexError("Calculation of GLSZM Failed.")
    if rc < 0:
        _lib.raise_for(rc, "GLSZM")
    maxRegion = max(int(rc), 1)
    out = torch.empty((Ng, maxRegion), dtype=torch.float64, device=image.device)
    rc = lib.prad_fill_glszm_dev(C.c_void_p(out.data_ptr()), 1, int(Ng), maxRegion, _stream_ptr())
    if rc == _lib.PRAD_INDEX_ERROR:
        raise IndexError("Error filling GLSZM.")
    _lib.raise_for(rc, "GLSZM")
    return out


def glszm_compact(image: torch.Tensor, mask: torch.Tensor, Ng: int, Ns: int | None = None, force2D: bool = False,
                  force2Ddimension: int = 0):
    """GLSZM without its empty size columns: (P float64 [Ng, k] on the device, sizes int32 numpy [k] ascending),
    P[:, c] counting the zones of sizes[c].  Equals glszm(...)[:, sizes - 1]; this is what the feature class keeps
    after glszm.py:118-131, without ever materialising the [Ng, maxRegion] layout."""
    lib, image, mask, size, f2d, angles = _neigh_common(image, mask, None, force2D, force2Ddimension)
    Na, Nd = angles.shape
    if Ns is None:
        Ns = int(mask.sum().item())
    nz = C.c_longlong(0)
    rc = lib.prad_calculate_glszm_dev(C.c_void_p(image.data_ptr()), C.c_void_p(mask.data_ptr()), _iptr(size), Nd,
                                      _iptr(angles), Na, int(Ng), int(Ns), 1, None, 0, f2d, C.byref(nz),
                                      _stream_ptr())
    if rc == _lib.PRAD_E_INDEX:
        raise IndexError("Calculation of GLSZM Failed.")
    if rc < 0:
        _lib.raise_for(rc, "GLSZM")
    maxRegion = max(int(rc), 1)
    cap = int(min(maxRegion, int(np.sqrt(2.0 * image.numel())) + 2))
    sizes = np.empty(cap, dtype=np.intc)
    k = lib.prad_glszm_sizes(_iptr(sizes), cap)
    if k < 0:
        _lib.raise_for(k, "GLSZM sizes")
    out = torch.empty((Ng, max(k, 1)), dtype=torch.float64, device=image.device)
    rc = lib.prad_fill_glszm_compact_dev(C.c_void_p(out.data_ptr()), int(Ng), int(k), _stream_ptr())
    if rc == _lib.PRAD_INDEX_ERROR:
        raise IndexError("Error filling GLSZM.")
    _lib.raise_for(rc, "GLSZM")
    return out[:, :k], sizes[:k].copy()


def glszm_features(image: torch.Tensor, mask: torch.Tensor, Ng: int, Ns: int | None = None, force2D: bool = False,
                   force2Ddimension: int = 0, deferred: bool = False):
    """the 16 GLSZM features of a segment without a host round trip between zone labelling, the compact matrix and the
    formulas (prad_glszm_features_dev): (float64 numpy [17], int32 numpy [1]) -- values, then the verdict (0 = fine, else
    take glszm_compact + zone_matrix_features, which raise what the reference raises); flag != 0 = no zone.
    deferred=True: enqueue only, both arrays live in the result arena and are valid after deferred_status().
    Raises NotImplementedError when the volume is not one for the packed-byte tile kernels."""
    lib, image, mask, size, f2d, angles = _neigh_common(image, mask, None, force2D, force2Ddimension)
    Na, Nd = angles.shape
    if Ns is None:
        Ns = int(mask.sum().item())
    alloc = result_array if deferred else np.empty
    out = alloc((17,), np.float64)
    empty = alloc((1,), np.intc)
    with _Deferred(lib, deferred):
        rc = lib.prad_glszm_features_dev(C.c_void_p(image.data_ptr()), C.c_void_p(mask.data_ptr()), _iptr(size), Nd,
                                         _iptr(angles), Na, int(Ng), int(Ns), out.ctypes.data_as(C.POINTER(C.c_double)),
                                         _iptr(empty), _stream_ptr())
    if rc == _lib.PRAD_E_INDEX:
        raise IndexError("Calculation of GLSZM Failed.")
    _lib.raise_for(rc, "GLSZM features")
    if deferred:
        _deferred_keep.append((image, mask))
    return out, empty


def voxel_glcm_features(image: torch.Tensor, mask: torch.Tensor, Ng: int, voxels: torch.Tensor, features,
                        kernelRadius: int = 1, force2D: bool = False, force2Ddimension: int = 0,
                        symmetrical: bool = True, distances=(1,)):
    """Fused voxel-based GLCM feature maps, everything device-resident.
    voxels: int32 [Nd, Nvox] centre coordinates (np.where layout) on the device.
    Returns {name: float64 tensor [Nvox]} (JointAverage with the reference's plain-mean NaN rule applied)."""
    from .cmatrices import VOXEL_GLCM_FEATURES
    lib, image, mask, size = _prep(image, mask)
    f2d = int(force2Ddimension) if force2D else -1
    angles = _build_angles(size, list(distances), False, f2d)
    Na, Nd = angles.shape
    vox = voxels.to(torch.int32).contiguous()
    if vox.dim() != 2 or vox.shape[0] != Nd:
        raise RuntimeError("Expecting voxel indices array to be 2-dimensional")
    Nvox = int(vox.shape[1])
    ids = np.array([VOXEL_GLCM_FEATURES.index(f) for f in features], dtype=np.intc)
    out = torch.empty((len(ids), Nvox), dtype=torch.float64, device=image.device)
    empty = torch.empty(Nvox, dtype=torch.int32, device=image.device)
    anyne = torch.zeros(1, dtype=torch.int32, device=image.device)
    rc = lib.prad_voxel_glcm_features_dev(
        C.c_void_p(image.data_ptr()), C.c_void_p(mask.data_ptr()), _iptr(size), Nd, _iptr(angles), Na, int(Ng), Nvox,
        C.c_void_p(vox.data_ptr()), int(kernelRadius), f2d, 1 if symmetrical else 0, _iptr(ids), len(ids),
        C.c_void_p(out.data_ptr()), C.c_void_p(empty.data_ptr()), C.c_void_p(anyne.data_ptr()), _stream_ptr())
    _lib.raise_for(rc, "voxel GLCM features")
    res = {f: out[i] for i, f in enumerate(features)}
    if "JointAverage" in res:
        bad = (empty & anyne[0]) != 0
        res["JointAverage"] = torch.where(bad, torch.full_like(res["JointAverage"], float("nan")), res["JointAverage"])
    return res


def voxel_texture_features(family: int, image: torch.Tensor, mask: torch.Tensor, Ng: int, voxels: torch.Tensor,
                           feature_ids, kernelRadius: int = 1, force2D: bool = False, force2Ddimension: int = 0,
                           distances=(1,), alpha: int = 0) -> torch.Tensor:
    """Fused voxel-based GLDM (family 1) / NGTDM (2) / GLRLM (3) / GLSZM (4) feature maps: float64 tensor
    [nfeat, Nvox] on the device.  voxels: int32 [Nd, Nvox] centre coordinates on the device."""
    lib, image, mask, size = _prep(image, mask)
    f2d = int(force2Ddimension) if force2D else -1
    # GLRLM walks one direction per line, the other three look at the full neighbourhood; GLRLM / GLSZM ignore
    # `distances` (_cmatrices.c:283,475)
    angles = _build_angles(size, list(distances) if family in (1, 2) else None, family != 3, f2d)
    Na, Nd = angles.shape
    vox = voxels.to(torch.int32).contiguous()
    if vox.dim() != 2 or vox.shape[0] != Nd:
        raise RuntimeError("Expecting voxel indices array to be 2-dimensional")
    ids = np.ascontiguousarray(feature_ids, dtype=np.intc)
    out = torch.empty((len(ids), int(vox.shape[1])), dtype=torch.float64, device=image.device)
    rc = lib.prad_voxel_texture_features_dev(
        int(family), C.c_void_p(image.data_ptr()), C.c_void_p(mask.data_ptr()), _iptr(size), Nd, _iptr(angles), Na,
        int(Ng), int(alpha), int(vox.shape[1]), C.c_void_p(vox.data_ptr()), int(kernelRadius), f2d, _iptr(ids), len(ids),
        C.c_void_p(out.data_ptr()), _stream_ptr())
    _lib.raise_for(rc, "voxel texture features")
    return out


# ---- device-resident discretisation and filters (config 3: filter stack -> re-binning -> matrices, all in HBM) ---
_DTYPE_CODES = {torch.float32: 0, torch.float64: 1, torch.int32: 2, torch.int16: 3}


def _mask_u8(mask: torch.Tensor) -> torch.Tensor:
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    return mask.contiguous() if mask.dtype == torch.uint8 else (mask != 0).view(torch.uint8)


def bin_image(image: torch.Tensor, mask: torch.Tensor, with_counts: bool = False, **kwargs):
    """imageoperations.binImage + base._applyBinning on the device: returns (levels int32 tensor, Ng, edges), plus the
    ROI voxel count per level (int64 numpy [Ng + 1], [0] = ROI voxels below the first edge: none) with `with_counts` --
    produced by the same pass over the image (prad_digitize_counts_dev).
    The edges come from pyradiomics_amd.imageoperations.getBinEdges fed with the ROI's (min, max), which is all
    that function depends on, so levels are identical to the host route."""
    from . import imageoperations
    lib = _lib.load()
    if image.dtype not in _DTYPE_CODES:
        image = image.to(torch.float64)
    image = image.contiguous()
    mask = _mask_u8(mask)
    rc = lib.prad_set_device(image.device.index or 0)
    mm = (C.c_double * 2)()
    n = image.numel()
    binCount = kwargs.get("binCount")
    # float32 images: the device builds np.histogram's edges in float32 arithmetic as NumPy >= 2 does (NEP 50); NumPy 1.x
    # computes them in float64 and casts, which can differ by one ulp -> the host-built edges serve those installations
    f32_edges_ok = image.dtype != torch.float32 or _NUMPY2
    if (binCount is not None and 1 <= int(binCount) <= 4096 and f32_edges_ok
            and not os.environ.get("PRAD_BIN_TWO_CALLS")):
        # fixed bin count: edges built on the device, one synchronisation instead of two
        nb = int(binCount)
        levels = torch.empty(image.shape, dtype=torch.int32, device=image.device)
        edges = np.empty(nb + 1, dtype=np.float64)
        counts = np.zeros(nb + 2, dtype=np.int64)
        top = C.c_int(0)
        rc = lib.prad_bincount_dev(C.c_void_p(image.data_ptr()), _DTYPE_CODES[image.dtype], C.c_void_p(mask.data_ptr()), n, nb,
                                       C.c_void_p(levels.data_ptr()), mm, edges.ctypes.data_as(C.POINTER(C.c_double)),
                                       C.byref(top), counts.ctypes.data_as(C.POINTER(C.c_longlong)), _stream_ptr())
        if rc != _lib.PRAD_E_UNSUPPORTED:
            _lib.raise_for(rc, "bincount")
            if with_counts:
                return levels, int(top.value), edges, counts[:int(top.value) + 1]
            return levels, int(top.value), edges
    rc = lib.prad_roi_minmax_dev(C.c_void_p(image.data_ptr()), _DTYPE_CODES[image.dtype], C.c_void_p(mask.data_ptr()),
                                 n, mm, _stream_ptr())
    _lib.raise_for(rc, "roi_minmax")
    np_dtype = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32, torch.int16: np.int16}[image.dtype]
    edges = np.asarray(imageoperations.getBinEdges(np.array([mm[0], mm[1]], dtype=np_dtype), **kwargs), dtype=np.float64)
    levels = torch.empty(image.shape, dtype=torch.int32, device=image.device)
    top = C.c_int(0)
    counts = np.zeros(len(edges) + 1, dtype=np.int64) if with_counts else None
    rc = lib.prad_digitize_counts_dev(C.c_void_p(image.data_ptr()), _DTYPE_CODES[image.dtype], C.c_void_p(mask.data_ptr()), n,
                                      edges.ctypes.data_as(C.POINTER(C.c_double)), len(edges), C.c_void_p(levels.data_ptr()),
                                      C.byref(top), counts.ctypes.data_as(C.POINTER(C.c_longlong)) if with_counts else None,
                                      _stream_ptr())
    _lib.raise_for(rc, "digitize")
    if with_counts:
        return levels, int(top.value), edges, counts[:int(top.value) + 1]
    return levels, int(top.value), edges


def bin_image_enqueue(image: torch.Tensor, mask: torch.Tensor, **kwargs):
    """the queueing half of bin_image for a fixed bin count (prad_bincount_enqueue_dev): the kernels and the copy of their few
    result words are queued on the current stream, nothing is waited for.  Returns a token for bin_image_collect, or None when
    the request needs the synchronous route (binWidth, a float32 image under NumPy 1.x, more than four binnings in flight)."""
    lib = _lib.load()
    binCount = kwargs.get("binCount")
    if binCount is None or not (1 <= int(binCount) <= 4096) or os.environ.get("PRAD_BIN_TWO_CALLS") or os.environ.get("PRAD_BIN_SYNC"):
        return None
    if image.dtype not in _DTYPE_CODES:
        image = image.to(torch.float64)
    if image.dtype == torch.float32 and not _NUMPY2:
        return None
    image = image.contiguous()
    mask8 = _mask_u8(mask)
    lib.prad_set_device(image.device.index or 0)
    levels = torch.empty(image.shape, dtype=torch.int32, device=image.device)
    ticket = C.c_int(-1)
    rc = lib.prad_bincount_enqueue_dev(C.c_void_p(image.data_ptr()), _DTYPE_CODES[image.dtype], C.c_void_p(mask8.data_ptr()),
                                       image.numel(), int(binCount), C.c_void_p(levels.data_ptr()), C.byref(ticket), _stream_ptr())
    if rc != _lib.PRAD_OK:
        return None
    return {"ticket": int(ticket.value), "levels": levels, "nb": int(binCount), "keep": (image, mask8), "args": (image, mask, kwargs)}


def bin_image_collect(token, with_counts: bool = False):
    """waits for a bin_image_enqueue() token and returns what bin_image returns (a constant or non-finite ROI takes the
    synchronous two-call route, as there)"""
    lib = _lib.load()
    nb = token["nb"]
    mm = (C.c_double * 2)()
    edges = np.empty(nb + 1, dtype=np.float64)
    counts = np.zeros(nb + 2, dtype=np.int64)
    top = C.c_int(0)
    rc = lib.prad_bincount_wait(int(token["ticket"]), mm, edges.ctypes.data_as(C.POINTER(C.c_double)), C.byref(top),
                                counts.ctypes.data_as(C.POINTER(C.c_longlong)))
    token["keep"] = None
    image, mask, kwargs = token["args"]
    token["args"] = None
    if rc == _lib.PRAD_E_UNSUPPORTED:
        return bin_image(image, mask, with_counts=with_counts, **kwargs)
    _lib.raise_for(rc, "bincount")
    levels = token["levels"]
    if with_counts:
        return levels, int(top.value), edges, counts[:int(top.value) + 1]
    return levels, int(top.value), edges


def level_counts(levels: torch.Tensor, mask: torch.Tensor, Ng: int) -> np.ndarray:
    """int64 [Ng + 1]: ROI voxels per grey level 1..Ng ([0] = ROI voxels with a level outside that range)"""
    lib = _lib.load()
    levels = levels.contiguous()
    mask = _mask_u8(mask)
    lib.prad_set_device(levels.device.index or 0)
    counts = np.zeros(int(Ng) + 1, dtype=np.int64)
    rc = lib.prad_level_counts_dev(C.c_void_p(levels.data_ptr()), C.c_void_p(mask.data_ptr()), levels.numel(), int(Ng),
                                   counts.ctypes.data_as(C.POINTER(C.c_longlong)), _stream_ptr())
    _lib.raise_for(rc, "level_counts")
    return counts


FIRSTORDER_FIELDS = ("Np", "Energy", "Minimum", "P10", "P25", "Median", "P75", "P90", "Maximum", "Mean", "MAD", "rMAD",
                     "m2", "m3", "m4")


def firstorder_stats(image: torch.Tensor, mask: torch.Tensor, voxelArrayShift: float = 0.0) -> dict:
    """the ROI statistics behind the first-order feature class (prad_firstorder_dev): {field: float}"""
    lib = _lib.load()
    if image.dtype not in _DTYPE_CODES:
        image = image.to(torch.float64)
    image = image.contiguous()
    mask = _mask_u8(mask)
    lib.prad_set_device(image.device.index or 0)
    out = (C.c_double * len(FIRSTORDER_FIELDS))()
    rc = lib.prad_firstorder_dev(C.c_void_p(image.data_ptr()), _DTYPE_CODES[image.dtype], C.c_void_p(mask.data_ptr()),
                                 image.numel(), float(voxelArrayShift), out, _stream_ptr())
    _lib.raise_for(rc, "firstorder")
    return dict(zip(FIRSTORDER_FIELDS, (float(v) for v in out)))


def firstorder_stats_queue(image: torch.Tensor, mask: torch.Tensor, roi_count: int, voxelArrayShift: float = 0.0,
                           deferred: bool = False) -> np.ndarray:
    """firstorder_stats without host round trips between the passes (prad_firstorder_queue_dev): float64 numpy [16] --
    the FIRSTORDER_FIELDS values, then a verdict (0 = fine, else call firstorder_stats).  deferred=True: enqueue only,
    the array lives in the result arena and is valid after deferred_status() / deferred_wait().  NotImplementedError for
    integer images and ROIs below 2^20 voxels (firstorder_stats serves them with its exact histogram / full sort)."""
    lib = _lib.load()
    if image.dtype not in _DTYPE_CODES:
        image = image.to(torch.float64)
    image = image.contiguous()
    mask = _mask_u8(mask)
    lib.prad_set_device(image.device.index or 0)
    out = result_array((16,), np.float64) if deferred else np.empty(16, dtype=np.float64)
    with _Deferred(lib, deferred):
        rc = lib.prad_firstorder_queue_dev(C.c_void_p(image.data_ptr()), _DTYPE_CODES[image.dtype],
                                           C.c_void_p(mask.data_ptr()), image.numel(), int(roi_count),
                                           float(voxelArrayShift), out.ctypes.data_as(C.POINTER(C.c_double)), _stream_ptr())
    _lib.raise_for(rc, "firstorder queue")
    if deferred:
        _deferred_keep.append((image, mask))
    return out


def voxel_firstorder(image: torch.Tensor, mask: torch.Tensor, levels, voxels: torch.Tensor, feature_ids,
                     kernelRadius: int = 1, force2D: bool = False, force2Ddimension: int = 0, bbsize=None,
                     voxelArrayShift: float = 0.0, voxelVolume: float = 1.0) -> torch.Tensor:
    """Voxel-based first-order feature maps (prad_voxel_firstorder_dev): float64 tensor [nfeat, Nvox] on the device.
    voxels: int32 [Nd, Nvox] centre coordinates on the device; levels: discretised image (or None)."""
    lib = _lib.load()
    if image.dtype not in _DTYPE_CODES:
        image = image.to(torch.float64)
    image = image.contiguous()
    mask = _mask_u8(mask)
    lib.prad_set_device(image.device.index or 0)
    size = np.array(image.shape, dtype=np.intc)
    vox = voxels.to(torch.int32).contiguous()
    if vox.dim() != 2 or vox.shape[0] != image.dim():
        raise RuntimeError("Expecting voxel indices array to be 2-dimensional")
    ids = np.ascontiguousarray(feature_ids, dtype=np.intc)
    bb = None if bbsize is None else np.ascontiguousarray(bbsize, dtype=np.intc)
    if levels is not None:
        levels = levels.to(torch.int32).contiguous()
    out = torch.empty((len(ids), int(vox.shape[1])), dtype=torch.float64, device=image.device)
    rc = lib.prad_voxel_firstorder_dev(
        C.c_void_p(image.data_ptr()), _DTYPE_CODES[image.dtype], C.c_void_p(mask.data_ptr()),
        C.c_void_p(levels.data_ptr()) if levels is not None else None, _iptr(size), image.dim(), int(vox.shape[1]),
        C.c_void_p(vox.data_ptr()), int(kernelRadius), int(force2Ddimension) if force2D else -1,
        _iptr(bb) if bb is not None else None, float(voxelArrayShift), float(voxelVolume), _iptr(ids), len(ids),
        C.c_void_p(out.data_ptr()), _stream_ptr())
    _lib.raise_for(rc, "voxel firstorder")
    return out


def swt_level1(data: torch.Tensor, lo: np.ndarray, hi: np.ndarray, axes) -> torch.Tensor:
    """pywt.swtn(level=1) on the device: float64 tensor [2^len(axes), *data.shape], sub-bands in key order.  The image goes
    in as it is when the fused 3-D kernel takes the call (widened to float64 while the planes are staged), as a float64 copy
    otherwise.  (pywt.swtn widens integer images the same way but keeps float32 images in float32: for those the sub-bands
    here carry more precision than the reference's -- stated in DESIGN.md section 7.)"""
    lib = _lib.load()
    lib.prad_set_device(data.device.index or 0)
    size = np.array(data.shape, dtype=np.intc)
    ax = np.array(axes, dtype=np.intc)
    lo = np.ascontiguousarray(lo, dtype=np.float64)
    hi = np.ascontiguousarray(hi, dtype=np.float64)
    out = torch.empty((1 << len(ax),) + tuple(data.shape), dtype=torch.float64, device=data.device)
    if data.dtype in _DTYPE_CODES and data.dtype != torch.float64:
        raw = data.contiguous()
        rc = lib.prad_swt_level1_any_dev(C.c_void_p(raw.data_ptr()), _DTYPE_CODES[raw.dtype], _iptr(size), raw.dim(),
                                         C.c_void_p(lo.ctypes.data), C.c_void_p(hi.ctypes.data), len(lo), _iptr(ax), len(ax),
                                         C.c_void_p(out.data_ptr()), _stream_ptr())
        if rc != _lib.PRAD_E_UNSUPPORTED:
            _lib.raise_for(rc, "swt")
            return out
    data = data.to(torch.float64).contiguous()
    rc = lib.prad_swt_level1_dev(C.c_void_p(data.data_ptr()), _iptr(size), data.dim(), C.c_void_p(lo.ctypes.data),
                                 C.c_void_p(hi.ctypes.data), len(lo), _iptr(ax), len(ax), C.c_void_p(out.data_ptr()),
                                 _stream_ptr())
    _lib.raise_for(rc, "swt")
    return out


def wavelet_images(image: torch.Tensor, wavelet="coif1"):
    """the 8 (2^Nd) level-1 sub-bands of getWaveletImage as {name: float64 tensor}, yield order of the reference"""
    from .filters import wavelet_filters
    lo, hi = wavelet_filters(wavelet)
    x = image if image.dtype in _DTYPE_CODES else image.to(torch.float64)
    shape = x.shape
    for d, s in enumerate(shape):          # np.pad(..., 'wrap') by one sample on odd axes
        if s % 2:
            x = torch.cat([x, x.narrow(d, 0, 1)], dim=d)
    axes = tuple(range(x.dim() - 1, -1, -1))
    sub = swt_level1(x, lo, hi, axes)
    crop = tuple(slice(0, s) for s in shape)
    keys = [""]
    for _ in axes:
        keys = [k + c for k in keys for c in "ad"]
    out = {}
    approx = None
    for i, k in enumerate(keys):
        name = "wavelet-" + k.replace("a", "L").replace("d", "H")
        if k == "a" * len(axes):
            approx = (name, sub[i][crop])
        else:
            out[name] = sub[i][crop]
    out[approx[0]] = approx[1]
    return out


def _log_real(image: torch.Tensor) -> torch.Tensor:
    """the type the filter reads and returns: float64 inputs stay float64, everything else is float32 (SimpleITK's real type
    of the input, imageoperations.py:824-830).  Between the passes every image is float32 (ITK's InternalRealType)."""
    return image.contiguous() if image.dtype == torch.float64 else image.to(torch.float32).contiguous()


def log_images(image: torch.Tensor, spacing_xyz, sigmas, normalize: bool = True) -> list:
    """log_image for a list of sigmas, up to 8 per launch sequence (prad_log_multi_dev[_f64]): the same bits per sigma, the
    sigmas share the GPU.  float64 images in -> float64 out, otherwise float32."""
    lib = _lib.load()
    x = _log_real(image)
    fn = lib.prad_log_multi_dev_f64 if x.dtype == torch.float64 else lib.prad_log_multi_dev
    lib.prad_set_device(x.device.index or 0)
    size = np.array(x.shape, dtype=np.intc)
    sp = np.array([float(s) for s in spacing_xyz][::-1], dtype=np.float64)
    outs = []
    sig = [float(s) for s in sigmas]
    for lo in range(0, len(sig), 8):
        part = np.array(sig[lo:lo + 8], dtype=np.float64)
        res = [torch.empty_like(x) for _ in part]
        ptrs = (C.c_void_p * len(res))(*[r.data_ptr() for r in res])
        rc = fn(C.c_void_p(x.data_ptr()), _iptr(size), x.dim(), C.c_void_p(sp.ctypes.data),
                C.c_void_p(part.ctypes.data), len(res), 1 if normalize else 0, ptrs, _stream_ptr())
        _lib.raise_for(rc, "LoG")
        outs.extend(res)
    return outs


def log_image(image: torch.Tensor, spacing_xyz, sigma: float, normalize: bool = True) -> torch.Tensor:
    """sitk.LaplacianRecursiveGaussianImageFilter on the device: float32 tensor (float64 for a float64 input)"""
    lib = _lib.load()
    x = _log_real(image)
    fn = lib.prad_log_dev_f64 if x.dtype == torch.float64 else lib.prad_log_dev
    lib.prad_set_device(x.device.index or 0)
    size = np.array(x.shape, dtype=np.intc)
    sp = np.array([float(s) for s in spacing_xyz][::-1], dtype=np.float64)
    out = torch.empty_like(x)
    rc = fn(C.c_void_p(x.data_ptr()), _iptr(size), x.dim(), C.c_void_p(sp.ctypes.data), float(sigma),
            1 if normalize else 0, C.c_void_p(out.data_ptr()), _stream_ptr())
    _lib.raise_for(rc, "LoG")
    return out
