"""The reference's own CPython module `_cmatrices` (radiomics/src/_cmatrices.c + cmatrices.c, compiled unmodified into
oracle/_ref/_cmatrices.so by oracle/Makefile) as the SECOND oracle: argument parsing, dtype coercion (_cmatrices.c:1023-1085),
voxel list / set_bb (:1087-1147), output shapes and the exception type of every failure (:219,372,421,566,714,864, :905-1113).

CPU tier: oracle/binding.py's restatement of those semantics (which all other CPU tests rely on) == the real module.
GPU tier: pyradiomics_amd.cmatrices (the drop-in operator module over the C ABI) == the real module, call by call."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def wrapper():
    from oracle import binding
    if not binding.have_ref_wrapper():
        import os
        if os.path.isdir("/root/reference/radiomics/src"):
            binding.build()
        if not binding.have_ref_wrapper():
            pytest.skip("oracle/_ref/_cmatrices.so not available")
    return binding.ref_wrapper()


def _cases():
    rng = np.random.default_rng(77)
    vol = rng.integers(1, 9, size=(9, 11, 13))
    msk = rng.random(vol.shape) < 0.8
    vox = np.array(np.nonzero(msk))[:, ::17]
    out = []
    # dtype / layout coercions of image and mask (try_parse_arrays): force-cast to int32 / bool, C-contiguous copies
    for img in (vol.astype(np.int64), vol.astype(np.float64) + 0.6, vol.astype(np.uint8), np.asfortranarray(vol),
                vol.astype(np.int16)[::-1], vol.astype(np.float32)):
        for mk in (msk, msk.astype(np.uint8), msk.astype(np.int32) * 3, msk.astype(np.float64), np.asfortranarray(msk)):
            if np.asarray(img).shape == np.asarray(mk).shape:
                out.append((img, mk))
    return vol, msk, vox, out[:: 3]


def _same(a, b):
    if isinstance(a, tuple):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)


def _close_ngtdm(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return (a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a[..., 0], b[..., 0])
            and np.array_equal(a[..., 2], b[..., 2]) and np.allclose(a[..., 1], b[..., 1], rtol=1e-12, atol=0))


def _calls(img, mk, vox=None):
    """(name, args) for every operator, segment mode and (when vox is given) voxel mode"""
    Ng, Nr, Ns = 8, int(max(np.asarray(img).shape)), int(np.count_nonzero(mk))
    tail = () if vox is None else (2, vox)
    for f2d, dim in ((False, 0), (True, 0), (True, 2)):
        yield "calculate_glcm", (img, mk, np.array([1]), Ng, f2d, dim) + tail
        yield "calculate_glcm", (img, mk, np.array([1, 2]), Ng, f2d, dim) + tail
        yield "calculate_glrlm", (img, mk, Ng, Nr, f2d, dim) + tail
        yield "calculate_glszm", (img, mk, Ng, Ns, f2d, dim) + tail
        yield "calculate_gldm", (img, mk, np.array([1]), Ng, 1, f2d, dim) + tail
        yield "calculate_ngtdm", (img, mk, np.array([1]), Ng, f2d, dim) + tail


def _bad_calls(vol, msk, vox):
    """calls the reference answers with an exception"""
    ones = np.array([1])
    yield "calculate_glcm", (vol, msk[:-1], ones, 8, False, 0)                      # ValueError: shapes differ
    yield "calculate_glcm", (vol, msk[0], ones, 8, False, 0)                        # ValueError: ranks differ
    yield "calculate_glrlm", (vol, msk[:, :-1], 8, 13, False, 0)
    yield "calculate_glcm", (vol, msk, ones, 4, False, 0)                           # IndexError: level > Ng under the mask
    yield "calculate_gldm", (vol, msk, ones, 4, 0, False, 0)
    yield "calculate_ngtdm", (vol, msk, ones, 4, False, 0)
    yield "calculate_glrlm", (vol, msk, 4, 13, False, 0)
    yield "calculate_glszm", (vol, msk, 4, int(msk.sum()), False, 0)
    yield "calculate_glcm", (vol * 0, msk, ones, 8, False, 0)                       # level 0 under the mask
    yield "calculate_glcm", (vol, msk, np.array([0]), 8, False, 0)                  # RuntimeError: distance < 1
    yield "calculate_glcm", (vol, msk, np.array([[1, 2]]), 8, False, 0)             # ValueError: distances not 1-D
    yield "calculate_glcm", (vol, msk, ones, 8, False, 0, 0, vox)                   # RuntimeError: voxels without kernelRadius
    yield "calculate_glcm", (vol, msk, ones, 8, False, 0, 2, vox[:2])               # RuntimeError: voxels not (Nd, Nvox)
    yield "calculate_glcm", (vol, msk, ones, 8, False, 0, 2, vox.ravel())
    yield "generate_angles", (np.array([[5, 5, 5]]), ones, True, False, 0)          # ValueError: size not 1-D
    yield "generate_angles", (np.array([5, 5, 5]), np.array([0]), True, False, 0)


def _compare(ours, theirs, vol, msk, vox, cases):
    n = 0
    for img, mk in cases:
        for name, args in _calls(img, mk):
            want = getattr(theirs, name)(*args)
            got = getattr(ours, name)(*args)
            ok = _close_ngtdm(got, want) if name == "calculate_ngtdm" else _same(got, want)
            assert ok, (name, np.asarray(img).dtype, np.asarray(mk).dtype)
            n += 1
    for name, args in _calls(vol, msk, vox):                                        # voxel mode: per-kernel bounding boxes
        want = getattr(theirs, name)(*args)
        got = getattr(ours, name)(*args)
        ok = _close_ngtdm(got, want) if name == "calculate_ngtdm" else _same(got, want)
        assert ok, (name, "voxel mode")
        n += 1
    for size, dist, bidir, f2d, dim in (((5, 6, 7), [1], True, False, 0), ((5, 6, 7), [1, 2], False, True, 1),
                                        ((1, 6, 7), [1], True, False, 0), ((3, 4), [1, 3], True, False, 0)):
        a = (np.array(size), np.array(dist), bidir, f2d, dim)
        assert _same(ours.generate_angles(*a), theirs.generate_angles(*a)), a
    for name, args in _bad_calls(vol, msk, vox):
        try:
            getattr(theirs, name)(*args)
        except Exception as e:                    # noqa: BLE001 -- whatever type the reference raises is the contract
            with pytest.raises(type(e)):
                getattr(ours, name)(*args)
            n += 1
        else:
            raise AssertionError("the reference accepted %s%r" % (name, tuple(type(a).__name__ for a in args)))
    return n


def test_binding_restatement_equals_the_reference_module(wrapper, oracle_ref):
    """oracle/binding.CMatricesCPU (ctypes over the reference's cmatrices.c + OUR reading of _cmatrices.c's wrapper) against
    the wrapper itself: pins the checker every other test uses"""
    vol, msk, vox, cases = _cases()
    assert _compare(oracle_ref, wrapper, vol, msk, vox, cases) > 100


def test_port_equals_the_reference_module(wrapper, oracle_port):
    vol, msk, vox, cases = _cases()
    assert _compare(oracle_port, wrapper, vol, msk, vox, cases[:3]) > 50


@pytest.mark.gpu
def test_drop_in_module_equals_the_reference_module(wrapper):
    """pyradiomics_amd.cmatrices -- what INTEGRATION.md binds as radiomics.cMatrices -- call by call against the module it
    replaces: same coercions, same output shapes / dtypes, same matrices, same exception types"""
    from pyradiomics_amd import cmatrices
    vol, msk, vox, cases = _cases()
    assert _compare(cmatrices, wrapper, vol, msk, vox, cases) > 100
