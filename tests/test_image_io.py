"""On-disk formats either side of the path: NRRD (what the reference's data/ uses), NIfTI-1 and MetaImage readers,
NRRD writer (feature maps)."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from pyradiomics_amd.image import Image, read_image, read_nrrd, write_nrrd

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_nrrd_roundtrip_and_reference_file(tmp_path):
    img = read_nrrd(os.path.join(GOLDEN, "data", "brain1_image.nrrd"))
    assert img.array.shape == (25, 256, 256) and img.array.dtype == np.int16 and img.GetSize() == (256, 256, 25)
    assert np.allclose(img.GetSpacing(), (0.7812499999999999, 0.7812499999999999, 6.499999999999998))
    for compress in (True, False):
        p = str(tmp_path / ("m%d.nrrd" % compress))
        sub = Image(img.array[3:9, 10:50, 20:41].astype(np.float64) * 0.5, img.spacing, (1.0, -2.0, 3.5), img.direction)
        write_nrrd(p, sub, compress=compress)
        back = read_image(p)
        assert np.array_equal(back.array, sub.array) and back.array.dtype == np.float64
        assert np.allclose(back.spacing, sub.spacing) and np.allclose(back.origin, sub.origin)
        assert np.allclose(back.direction, sub.direction)


def _nifti_bytes(arr, pixdim, sform=None, slope=0.0, inter=0.0, big=False):
    o = ">" if big else "<"
    codes = {"u1": 2, "i2": 4, "i4": 8, "f4": 16, "f8": 64, "u2": 512}
    hdr = bytearray(352)
    struct.pack_into(o + "i", hdr, 0, 348)
    dims = [arr.ndim] + list(arr.shape[::-1]) + [1] * (7 - arr.ndim)
    struct.pack_into(o + "8h", hdr, 40, *dims)
    struct.pack_into(o + "h", hdr, 70, codes[arr.dtype.str[1:]])
    struct.pack_into(o + "h", hdr, 72, arr.dtype.itemsize * 8)
    struct.pack_into(o + "8f", hdr, 76, 1.0, *pixdim, *([1.0] * (7 - len(pixdim))))
    struct.pack_into(o + "f", hdr, 108, 352.0)
    struct.pack_into(o + "2f", hdr, 112, slope, inter)
    if sform is not None:
        struct.pack_into(o + "2h", hdr, 252, 0, 1)
        struct.pack_into(o + "12f", hdr, 280, *np.asarray(sform, dtype=np.float64).ravel())
    hdr[344:348] = b"n+1\0"
    return bytes(hdr) + arr.astype(arr.dtype.newbyteorder(o)).tobytes()


def test_nifti_reader(tmp_path):
    rng = np.random.default_rng(0)
    arr = rng.integers(-500, 500, (5, 7, 9)).astype(np.int16)
    p = tmp_path / "a.nii"
    p.write_bytes(_nifti_bytes(arr, (0.5, 0.75, 2.0)))
    img = read_image(str(p))
    assert np.array_equal(img.array, arr) and img.array.dtype == np.int16 and np.allclose(img.spacing, (0.5, 0.75, 2.0))
    # gz + big endian + sform (RAS) + intensity scaling
    sform = [[-0.5, 0, 0, 10], [0, 0.75, 0, -20], [0, 0, 2.0, 30]]
    pz = tmp_path / "b.nii.gz"
    pz.write_bytes(gzip.compress(_nifti_bytes(arr, (0.5, 0.75, 2.0), sform, slope=2.0, inter=-1.0, big=True)))
    img = read_image(str(pz))
    assert np.array_equal(img.array, arr.astype(np.float64) * 2 - 1)
    assert np.allclose(img.spacing, (0.5, 0.75, 2.0)) and np.allclose(img.origin, (-10, 20, 30))
    assert np.allclose(np.array(img.direction).reshape(3, 3), np.diag([1.0, -1.0, 1.0]))     # RAS -> LPS
    with pytest.raises(ValueError):
        (tmp_path / "c.nii").write_bytes(b"\0" * 400)
        read_image(str(tmp_path / "c.nii"))


def test_metaimage_reader(tmp_path):
    rng = np.random.default_rng(1)
    arr = rng.standard_normal((4, 6, 8)).astype(np.float32)
    head = ("ObjectType = Image\nNDims = 3\nBinaryData = True\nBinaryDataByteOrderMSB = False\nCompressedData = %s\n"
            "TransformMatrix = 1 0 0 0 1 0 0 0 1\nOffset = 1.5 2.5 -3\nElementSpacing = 0.9 0.9 3\nDimSize = 8 6 4\n"
            "ElementType = MET_FLOAT\nElementDataFile = %s\n")
    (tmp_path / "a.mha").write_bytes((head % ("False", "LOCAL")).encode() + arr.tobytes())
    img = read_image(str(tmp_path / "a.mha"))
    assert np.array_equal(img.array, arr) and np.allclose(img.spacing, (0.9, 0.9, 3)) and np.allclose(img.origin, (1.5, 2.5, -3))
    (tmp_path / "b.zraw").write_bytes(zlib.compress(arr.tobytes()))
    (tmp_path / "b.mhd").write_text(head % ("True", "b.zraw"))
    assert np.array_equal(read_image(str(tmp_path / "b.mhd")).array, arr)
