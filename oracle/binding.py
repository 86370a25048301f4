"""ctypes driver for the CPU checkers (oracle/libtexture_oracle.so and oracle/_ref/libcmatrices_ref.so).

TEST INFRASTRUCTURE ONLY -- never imported by the product package (pyradiomics_amd).  Allowed
importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.

Both shared objects export the eight prototypes of the reference's radiomics/src/cmatrices.h:1-8.
`CMatricesCPU` puts the behaviour of the reference's CPython wrapper (radiomics/src/_cmatrices.c)
on top of them: dtype coercion (_cmatrices.c:1023-1085), angle construction (:926-1021), per-voxel
bounding boxes (:1120-1147), output shapes and the exception types of each failure.  It is written
independently of pyradiomics_amd/cmatrices.py on purpose: the checker shares no code with the
checked.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(_HERE, "libtexture_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libcmatrices_ref.so")

_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)


def build(verbose: bool = False) -> None:
    """Compile the restatement and, when /root/reference is present, the reference itself."""
    out = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)


def have_ref() -> bool:
    return os.path.exists(REF_SO)


REF_WRAPPER_SO = os.path.join(_HERE, "_ref", "_cmatrices.so")


def have_ref_wrapper() -> bool:
    return os.path.exists(REF_WRAPPER_SO)


def ref_wrapper():
    """The reference's own CPython extension module `_cmatrices` (radiomics/src/_cmatrices.c + cmatrices.c compiled unmodified
    by oracle/Makefile): exactly the object the reference binds as `radiomics.cMatrices` -- argument parsing, dtype coercion,
    set_bb, exception mapping and output allocation included."""
    import importlib.machinery
    import importlib.util
    loader = importlib.machinery.ExtensionFileLoader("_cmatrices", REF_WRAPPER_SO)
    spec = importlib.util.spec_from_loader("_cmatrices", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def _i(a):
    return a.ctypes.data_as(_ip)


class CMatricesCPU:
    """Python face of a cmatrices.h-compatible shared object (same call signatures as
    `radiomics._cmatrices`, _cmatrices.c:41-50)."""

    def __init__(self, so_path: str):
        if not os.path.exists(so_path):
            raise FileNotFoundError(so_path + " (run `make -C oracle`)")
        self.path = so_path
        L = C.CDLL(so_path)
        std = [_ip, C.c_char_p, _ip, _ip, _ip, _ip, C.c_int, C.c_int]
        L.calculate_glcm.argtypes = std + [_dp, C.c_int]
        L.calculate_glrlm.argtypes = std + [_dp, C.c_int, C.c_int]
        L.calculate_ngtdm.argtypes = std + [_dp, C.c_int]
        L.calculate_gldm.argtypes = std + [_dp, C.c_int, C.c_int]
        L.calculate_glszm.argtypes = std + [_ip, C.c_int, C.c_int, C.c_int]
        L.fill_glszm.argtypes = [_ip, _dp, C.c_int, C.c_int]
        L.get_angle_count.argtypes = [_ip, _ip, C.c_int, C.c_int, C.c_char, C.c_int]
        L.build_angles.argtypes = [_ip, _ip, C.c_int, C.c_int, C.c_int, C.c_int, _ip]
        for f in ("calculate_glcm", "calculate_glrlm", "calculate_ngtdm", "calculate_gldm",
                  "calculate_glszm", "fill_glszm", "get_angle_count", "build_angles"):
            getattr(L, f).restype = C.c_int
        self.L = L

    # -- first-order statistics: the reference computes these in numpy, so does this CPU backend ---------
    @staticmethod
    def firstorder_stats(image, mask, voxelArrayShift=0.0):
        from . import firstorder_oracle
        return firstorder_oracle.firstorder_stats(image, mask, voxelArrayShift)

    @staticmethod
    def voxel_firstorder(image, mask, levels, voxels, kernelRadius, bbsize, force2D, force2Ddimension,
                         voxelArrayShift, voxelVolume, features):
        from . import firstorder_oracle
        return firstorder_oracle.voxel_firstorder(image, mask, levels, voxels, kernelRadius, bbsize, force2D,
                                                  force2Ddimension, voxelArrayShift, voxelVolume, features)

    # -- helpers ---------------------------------------------------------------------------
    @staticmethod
    def _arrays(image, mask, copy_mask=False):
        img = np.ascontiguousarray(np.asarray(image).astype(np.intc, copy=False))
        msk = np.asarray(mask).astype(np.bool_, copy=False)
        msk = np.array(msk, order="C", copy=True) if copy_mask else np.ascontiguousarray(msk)
        if img.ndim != msk.ndim:
            raise ValueError("Expected image and mask to have equal number of dimensions.")
        if img.shape != msk.shape:
            raise ValueError("Dimensions of image and mask do not match.")
        size = np.array(img.shape, dtype=np.intc)
        strides = np.array([s // img.itemsize for s in img.strides], dtype=np.intc)
        return img, msk, size, strides

    def _angles(self, size, distances, bidirectional, force2Ddim):
        if distances is None:
            dist = np.array([1], dtype=np.intc)
        else:
            dist = np.ascontiguousarray(np.asarray(distances).astype(np.intc, copy=False))
            if dist.ndim != 1:
                raise ValueError("Expecting distances array to be 1-dimensional.")
        Nd = len(size)
        Na = self.L.get_angle_count(_i(size), _i(dist), Nd, len(dist), bytes([1 if bidirectional else 0]), force2Ddim)
        if Na == 0:
            raise RuntimeError("Error getting angle count.")
        ang = np.empty((Na, Nd), dtype=np.intc)
        if self.L.build_angles(_i(size), _i(dist), Nd, len(dist), force2Ddim, Na, _i(ang)) > 0:
            raise RuntimeError("Error building angles.")
        return ang

    @staticmethod
    def _voxels(voxels, Nd, kernelRadius):
        if voxels is None:
            return None, 1
        if kernelRadius <= 0:
            raise RuntimeError("Expecting kernelRadius > 0")
        v = np.ascontiguousarray(np.asarray(voxels).astype(np.intc, copy=False))
        if v.ndim != 2 or v.shape[0] != Nd:
            raise RuntimeError("Expecting voxel indices array to be 2-dimensional")
        return v, v.shape[1]

    @staticmethod
    def _bb(v, size, voxels, kernelRadius, force2Ddim):
        Nd = len(size)
        bb = np.zeros(2 * Nd, dtype=np.intc)
        if voxels is None:
            bb[Nd:] = size - 1
            return bb
        for d in range(Nd):
            c = int(voxels[d, v])
            if d == force2Ddim:
                bb[d] = bb[Nd + d] = c
            else:
                bb[d] = max(c - kernelRadius, 0)
                bb[Nd + d] = min(c + kernelRadius, int(size[d]) - 1)
        return bb

    def _run(self, fn, what, image, mask, distances, bidirectional, force2D, force2Ddim,
             kernelRadius, voxels, out_tail, extra):
        img, msk, size, strides = self._arrays(image, mask)
        vox, Nvox = self._voxels(voxels, img.ndim, kernelRadius)
        if not force2D:
            force2Ddim = -1
        ang = self._angles(size, distances, bidirectional, force2Ddim)
        Na, Nd = ang.shape
        tail = out_tail(Na)
        out = np.zeros((Nvox,) + tail, dtype=np.float64)
        per = int(np.prod(tail))
        flat = out.reshape(-1)
        for v in range(Nvox):
            bb = self._bb(v, size, vox, kernelRadius, force2Ddim)
            ok = fn(_i(img), msk.ctypes.data_as(C.c_char_p), _i(size), _i(bb), _i(strides), _i(ang), Na, Nd,
                    flat[v * per:(v + 1) * per].ctypes.data_as(_dp), *extra(Na))
            if not ok:
                raise IndexError("Calculation of %s Failed." % what)
        return out, ang

    # -- public API (signatures of _cmatrices.c) ---------------------------------------------
    def calculate_glcm(self, image, mask, distances, Ng, force2D, force2Ddimension, kernelRadius=0, voxels=None):
        return self._run(self.L.calculate_glcm, "GLCM", image, mask, distances, False, force2D, force2Ddimension,
                         kernelRadius, voxels, lambda Na: (Ng, Ng, Na), lambda Na: (Ng,))

    def calculate_glrlm(self, image, mask, Ng, Nr, force2D, force2Ddimension, kernelRadius=0, voxels=None):
        return self._run(self.L.calculate_glrlm, "GLRLM", image, mask, None, False, force2D, force2Ddimension,
                         kernelRadius, voxels, lambda Na: (Ng, Nr, Na), lambda Na: (Ng, Nr))

    def calculate_gldm(self, image, mask, distances, Ng, alpha, force2D, force2Ddimension, kernelRadius=0, voxels=None):
        return self._run(self.L.calculate_gldm, "GLDM", image, mask, distances, True, force2D, force2Ddimension,
                         kernelRadius, voxels, lambda Na: (Ng, 2 * Na + 1), lambda Na: (Ng, alpha))[0]

    def calculate_ngtdm(self, image, mask, distances, Ng, force2D, force2Ddimension, kernelRadius=0, voxels=None):
        return self._run(self.L.calculate_ngtdm, "NGTDM", image, mask, distances, True, force2D, force2Ddimension,
                         kernelRadius, voxels, lambda Na: (Ng, 3), lambda Na: (Ng,))[0]

    def calculate_glszm(self, image, mask, Ng, Ns, force2D, force2Ddimension, kernelRadius=0, voxels=None):
        img, msk, size, strides = self._arrays(image, mask, copy_mask=True)
        vox, Nvox = self._voxels(voxels, img.ndim, kernelRadius)
        if not force2D:
            force2Ddimension = -1
        ang = self._angles(size, None, True, force2Ddimension)
        Na, Nd = ang.shape
        if vox is not None:
            n = Nd - 1 if force2D else Nd
            Ns = min(Ns, (2 * kernelRadius + 1) ** n)
        temp = np.empty((Nvox, 2 * Ns + 1), dtype=np.intc)
        maxRegion = 0
        for v in range(Nvox):
            bb = self._bb(v, size, vox, kernelRadius, force2Ddimension)
            r = self.L.calculate_glszm(_i(img), msk.ctypes.data_as(C.c_char_p), _i(size), _i(bb), _i(strides),
                                       _i(ang), Na, Nd, _i(temp[v]), Ng, Ns, Nvox)
            if r < 0:
                raise IndexError("Calculation of GLSZM Failed.")
            maxRegion = max(maxRegion, r)
        maxRegion = max(maxRegion, 1)
        out = np.zeros((Nvox, Ng, maxRegion), dtype=np.float64)
        for v in range(Nvox):
            if not self.L.fill_glszm(_i(temp[v]), out[v].ctypes.data_as(_dp), Ng, maxRegion):
                raise IndexError("Error filling GLSZM.")
        return out

    def pairs_or_runs_for_angles(self, what, image, mask, Ng, Nr, angles):
        """calculate_glcm ("glcm") / calculate_glrlm ("glrlm") of the reference's core for an explicit angle list
        (the core takes the angle table as an argument, cmatrices.h:1-8) -> [Ng, Ng|Nr, na]"""
        img, msk, size, strides = self._arrays(image, mask)
        ang = np.ascontiguousarray(np.asarray(angles, dtype=np.intc))
        Na, Nd = ang.shape
        bb = self._bb(0, size, None, 0, -1)
        out = np.zeros((Ng, Ng if what == "glcm" else Nr, Na), dtype=np.float64)
        fn = self.L.calculate_glcm if what == "glcm" else self.L.calculate_glrlm
        extra = (Ng,) if what == "glcm" else (Ng, Nr)
        if not fn(_i(img), msk.ctypes.data_as(C.c_char_p), _i(size), _i(bb), _i(strides), _i(ang), Na, Nd,
                  out.ctypes.data_as(_dp), *extra):
            raise IndexError("Calculation of %s Failed." % what.upper())
        return out

    def glcm_glrlm_angle_sharded(self, image, mask, Ng, Nr, threads=None):
        """calculate_glcm + calculate_glrlm of a whole segment-mode volume with the 13 (or 4) unidirectional angles dealt to
        `threads` host threads, one core call per (matrix, angle): the core takes the angle table as an argument
        (cmatrices.c:4-92, :299-541), every angle writes its own [.., a] column, so the result equals the one-call matrices
        bit for bit (tests/test_oracle.py checks that) while a 512^3 volume takes seconds instead of half a minute.
        ctypes releases the GIL for the duration of a foreign call.  -> (glcm [Ng,Ng,Na], glrlm [Ng,Nr,Na], angles,
        {"wall_s", "cpu_s": sum of the per-call times, "threads"})"""
        import time
        from concurrent.futures import ThreadPoolExecutor
        img, msk, size, strides = self._arrays(image, mask)
        ang = self._angles(size, None, False, -1)
        Na, Nd = ang.shape
        bb = self._bb(0, size, None, 0, -1)
        if threads is None:
            threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        glcm = np.zeros((Ng, Ng, Na), dtype=np.float64)
        glrlm = np.zeros((Ng, Nr, Na), dtype=np.float64)

        def one(job):
            what, a = job
            one_angle = np.ascontiguousarray(ang[a:a + 1])
            out = np.zeros((Ng, Ng if what == "glcm" else Nr, 1), dtype=np.float64)
            fn = self.L.calculate_glcm if what == "glcm" else self.L.calculate_glrlm
            extra = (Ng,) if what == "glcm" else (Ng, Nr)
            t = time.perf_counter()
            ok = fn(_i(img), msk.ctypes.data_as(C.c_char_p), _i(size), _i(bb), _i(strides), _i(one_angle), 1, Nd,
                    out.ctypes.data_as(_dp), *extra)
            if not ok:
                raise IndexError("Calculation of %s Failed." % what.upper())
            (glcm if what == "glcm" else glrlm)[:, :, a] = out[:, :, 0]
            return time.perf_counter() - t
        jobs = [("glrlm", a) for a in range(Na)] + [("glcm", a) for a in range(Na)]       # the slower calls first
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=max(1, min(threads, len(jobs)))) as pool:
            cpu_s = sum(pool.map(one, jobs))
        wall = time.perf_counter() - t0
        # cmatrices.c:524-534 per angle already handled inside each one-angle call
        return glcm, glrlm, ang, {"wall_s": wall, "cpu_s": cpu_s, "threads": max(1, min(threads, len(jobs)))}

    def generate_angles(self, size, distances, bidirectional, force2D, force2Ddimension):
        size = np.ascontiguousarray(np.asarray(size).astype(np.intc, copy=False))
        if size.ndim != 1:
            raise ValueError("Expected a 1D array for size")
        return self._angles(size, distances, bool(bidirectional), force2Ddimension if force2D else -1)


def port() -> CMatricesCPU:
    return CMatricesCPU(PORT_SO)


def ref() -> CMatricesCPU:
    return CMatricesCPU(REF_SO)
