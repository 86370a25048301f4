"""Minimal volume container + NRRD I/O standing in for the few `SimpleITK.Image` services the hot path's callers
use (radiomics/base.py:84-96 GetArrayFromImage, glcm.py:163 GetSpacing, featureextractor.py:399-483 ReadImage).
SimpleITK is not a dependency of this package; arrays are numpy (z, y, x), spacing is (x, y, z) as in SimpleITK."""
from __future__ import annotations

import gzip
import os

import numpy as np

_NRRD_TYPES = {
    "signed char": "i1", "int8": "i1", "int8_t": "i1", "uchar": "u1", "unsigned char": "u1", "uint8": "u1",
    "uint8_t": "u1", "short": "i2", "short int": "i2", "signed short": "i2", "int16": "i2", "int16_t": "i2",
    "ushort": "u2", "unsigned short": "u2", "uint16": "u2", "uint16_t": "u2", "int": "i4", "signed int": "i4",
    "int32": "i4", "int32_t": "i4", "uint": "u4", "unsigned int": "u4", "uint32": "u4", "uint32_t": "u4",
    "longlong": "i8", "long long": "i8", "int64": "i8", "int64_t": "i8", "ulonglong": "u8",
    "unsigned long long": "u8", "uint64": "u8", "uint64_t": "u8", "float": "f4", "double": "f8",
}


class Image:
    """A volume with SimpleITK-style geometry.  Storage is a numpy array (z, y, x) on the host, a torch tensor in
    HBM, or both: `array` downloads on first use, `device_tensor()` uploads on first use, so a derived image made
    on the device (filter output, crop) never touches the host unless somebody asks for `.array`.
    spacing/origin: (x, y, z) tuples; direction: Nd*Nd floats row-major."""

    def __init__(self, array=None, spacing=None, origin=None, direction=None, tensor=None):
        if array is None and tensor is None:
            raise ValueError("Image needs an array or a device tensor")
        self._array = None if array is None else np.asarray(array)
        self._tensor = tensor
        nd = len(self.shape)
        self.spacing = tuple(float(s) for s in (spacing if spacing is not None else (1.0,) * nd))
        self.origin = tuple(float(s) for s in (origin if origin is not None else (0.0,) * nd))
        self.direction = tuple(direction) if direction is not None else tuple(np.eye(nd).ravel())
        self._derived = {}           # per-image memo of device-side products (ROI, bounding box, levels)

    @property
    def array(self):
        if self._array is None:
            self._array = self._tensor.cpu().numpy()
        return self._array

    @property
    def shape(self):
        return tuple(self._array.shape) if self._array is not None else tuple(self._tensor.shape)

    @property
    def on_device(self):
        return self._tensor is not None

    def device_tensor(self, device=None):
        """the volume as a torch tensor in HBM (uploaded once, original dtype; 64-bit integers narrow to int32
        as the operator boundary does, _cmatrices.c:1027)"""
        import torch
        if self._tensor is None:
            a = np.ascontiguousarray(self._array)
            if a.dtype in (np.uint16, np.uint32, np.uint64, np.int64):
                a = a.astype(np.int32) if (a.size == 0 or (a.min() >= -2**31 and a.max() < 2**31)) else a.astype(np.float64)
            elif a.dtype == np.uint8 or a.dtype == np.int8:
                a = a.astype(np.int16)
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
            self._tensor = torch.from_numpy(a).to(dev, non_blocking=False)
        return self._tensor

    # the SimpleITK-style accessors the feature classes call
    def GetSpacing(self):
        return self.spacing

    def GetOrigin(self):
        return self.origin

    def GetDirection(self):
        return self.direction

    def GetSize(self):
        return tuple(int(s) for s in self.shape[::-1])

    def GetDimension(self):
        return len(self.shape)

    def like(self, array=None, tensor=None):
        """new Image with this one's geometry (sitk CopyInformation, base.py:240-245)"""
        return Image(array, self.spacing, self.origin, self.direction, tensor=tensor)


def as_image(obj):
    """Image view of an Image / array-like / SimpleITK-like object (geometry kept where available)"""
    if isinstance(obj, Image):
        return obj
    if hasattr(obj, "GetSpacing") and not isinstance(obj, np.ndarray):
        return Image(as_array(obj), obj.GetSpacing(), obj.GetOrigin(), obj.GetDirection())
    if hasattr(obj, "data_ptr"):
        return Image(tensor=obj)
    return Image(np.asarray(obj))


def as_array(obj):
    """numpy view of an Image / array-like / anything exposing a SimpleITK-like interface."""
    if isinstance(obj, Image):
        return obj.array
    if hasattr(obj, "GetSpacing") and not isinstance(obj, np.ndarray):   # a real SimpleITK image
        import SimpleITK as sitk  # pragma: no cover - optional
        return sitk.GetArrayFromImage(obj)
    if hasattr(obj, "data_ptr"):
        return obj.cpu().numpy()
    return np.asarray(obj)


def spacing_of(obj):
    if hasattr(obj, "GetSpacing"):
        return tuple(obj.GetSpacing())
    return (1.0,) * np.asarray(obj).ndim


def read_nrrd(path: str) -> Image:
    """NRRD0004/5 reader for attached-data files with raw or gzip encoding (what data/*.nrrd use)."""
    with open(path, "rb") as f:
        raw = f.read()
    if not raw.startswith(b"NRRD"):
        raise ValueError("%s: not an NRRD file" % path)
    sep = raw.find(b"\n\n")
    crlf = raw.find(b"\r\n\r\n")
    if crlf != -1 and (sep == -1 or crlf < sep):
        header, data = raw[:crlf], raw[crlf + 4:]
    elif sep != -1:
        header, data = raw[:sep], raw[sep + 2:]
    else:
        raise ValueError("%s: NRRD header is not terminated" % path)
    fields = {}
    for line in header.decode("ascii", "replace").splitlines()[1:]:
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        if ":=" in line:
            continue                       # key/value metadata
        if ":" in line:
            k, v = line.split(":", 1)
            fields[k.strip().lower()] = v.strip()
    if "data file" in fields or "datafile" in fields:
        raise NotImplementedError("detached NRRD data files are not supported")
    dt = _NRRD_TYPES.get(fields["type"].lower())
    if dt is None:
        raise NotImplementedError("NRRD type %r" % fields["type"])
    sizes = [int(s) for s in fields["sizes"].split()]
    enc = fields.get("encoding", "raw").lower()
    if enc in ("gzip", "gz"):
        data = gzip.decompress(data)
    elif enc != "raw":
        raise NotImplementedError("NRRD encoding %r" % enc)
    order = "<" if fields.get("endian", "little").lower() == "little" else ">"
    dtype = np.dtype(order + dt) if dt[1] != "1" else np.dtype(dt)
    n = int(np.prod(sizes))
    arr = np.frombuffer(data, dtype=dtype, count=n).reshape(sizes[::-1])
    arr = arr.astype(dtype.newbyteorder("="), copy=True)
    nd = len(sizes)
    # a segmentation object stored as a vector image: the first (fastest) axis carries the components, e.g.
    # "kinds: list domain domain domain" / "space directions: none (..) (..) (..)" -> array (z, y, x, c), geometry of
    # the remaining axes (imageoperations.getMask picks the channel, imageoperations.py:12-64)
    kinds = fields.get("kinds", "").lower().split()
    sdirs = fields.get("space directions", "").lower().split()
    vector = nd >= 2 and ((len(kinds) == nd and kinds[0] not in ("domain", "space")) or (sdirs[:1] == ["none"]))
    if vector:
        nd -= 1
    spacing, direction = [1.0] * nd, list(np.eye(nd).ravel())
    if "space directions" in fields:
        vecs = [v for v in fields["space directions"].replace("none", "").split(")") if "(" in v]
        for i, v in enumerate(vecs[:nd]):
            comp = [float(x) for x in v[v.index("(") + 1:].split(",")]
            norm = float(np.sqrt(sum(c * c for c in comp)))
            spacing[i] = norm
            for r in range(min(nd, len(comp))):
                direction[r * nd + i] = comp[r] / norm if norm else 0.0
    elif "spacings" in fields:
        spacing = [float(s) for s in fields["spacings"].split() if s.lower() != "nan"][-nd:]
    origin = [0.0] * nd
    if "space origin" in fields:
        o = fields["space origin"]
        origin = [float(x) for x in o[o.index("(") + 1:o.index(")")].split(",")]
    img = Image(arr, spacing, origin, direction)
    if vector:
        img.components = int(sizes[0])
    return img


def write_nrrd(path: str, image: Image, compress: bool = True) -> None:
    """Writes feature maps / masks the way scripts/voxel.py:67-73 does through SimpleITK (gzip NRRD)."""
    arr = np.ascontiguousarray(image.array)
    rev = {v: k for k, v in (("short", "i2"), ("ushort", "u2"), ("int", "i4"), ("uint", "u4"), ("uchar", "u1"),
                             ("signed char", "i1"), ("float", "f4"), ("double", "f8"), ("longlong", "i8"),
                             ("ulonglong", "u8"))}
    key = arr.dtype.kind + str(arr.dtype.itemsize)
    if key not in rev:
        raise NotImplementedError("dtype %s" % arr.dtype)
    nd = arr.ndim
    d = np.array(image.direction, dtype=float).reshape(nd, nd)
    dirs = " ".join("(" + ",".join(repr(float(d[r, i] * image.spacing[i])) for r in range(nd)) + ")" for i in range(nd))
    hdr = ["NRRD0004", "type: " + rev[key], "dimension: %d" % nd, "space dimension: %d" % nd,
           "sizes: " + " ".join(str(s) for s in arr.shape[::-1]), "space directions: " + dirs,
           "kinds: " + " ".join(["domain"] * nd), "endian: little", "encoding: " + ("gzip" if compress else "raw"),
           "space origin: (" + ",".join(repr(float(o)) for o in image.origin) + ")"]
    payload = arr.astype(arr.dtype.newbyteorder("<"), copy=False).tobytes()
    if compress:
        payload = gzip.compress(payload, 6)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(("\n".join(hdr) + "\n\n").encode("ascii"))
        f.write(payload)
    os.replace(tmp, path)


# ---- other on-disk formats the reference reads through SimpleITK (featureextractor.py:399-483 ReadImage) ----------
_NIFTI_TYPES = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4", 1024: "i8", 1280: "u8"}


def read_nifti(path: str) -> Image:
    """NIfTI-1 single-file reader (.nii / .nii.gz): voxel array as (z, y, x), spacing from pixdim, origin / direction
    from the sform or qform converted from NIfTI's RAS to the LPS convention SimpleITK reports; scl_slope / scl_inter
    applied like ITK does (result float64 when a scaling is present)."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    order = "<" if int(np.frombuffer(raw, "<i4", 1, 0)[0]) == 348 else ">"
    if int(np.frombuffer(raw, order + "i4", 1, 0)[0]) != 348 or raw[344:347] not in (b"n+1", b"ni1"):
        raise ValueError("%s: not a NIfTI-1 file" % path)
    if raw[344:347] == b"ni1":
        raise NotImplementedError("NIfTI header/image pairs (.hdr/.img) are not supported")
    dim = np.frombuffer(raw, order + "i2", 8, 40)
    nd = int(dim[0])
    if nd < 2 or nd > 3 and any(int(d) > 1 for d in dim[4:nd + 1]):
        raise NotImplementedError("only 2-D and 3-D NIfTI volumes are supported (dim = %s)" % list(dim))
    nd = min(nd, 3)
    sizes = [int(d) for d in dim[1:nd + 1]]
    datatype = int(np.frombuffer(raw, order + "i2", 1, 70)[0])
    if datatype not in _NIFTI_TYPES:
        raise NotImplementedError("NIfTI datatype %d" % datatype)
    pixdim = np.frombuffer(raw, order + "f4", 8, 76)
    vox_offset = int(np.frombuffer(raw, order + "f4", 1, 108)[0])
    slope, inter = (float(v) for v in np.frombuffer(raw, order + "f4", 2, 112))
    dt = np.dtype(_NIFTI_TYPES[datatype])
    if dt.itemsize > 1:
        dt = dt.newbyteorder(order)
    arr = np.frombuffer(raw, dt, int(np.prod(sizes)), vox_offset).reshape(sizes[::-1])
    arr = arr.astype(dt.newbyteorder("="), copy=True)
    if slope not in (0.0, 1.0) or (slope != 0.0 and inter != 0.0):
        arr = arr.astype(np.float64) * slope + inter
    spacing = [abs(float(p)) or 1.0 for p in pixdim[1:nd + 1]]
    origin, direction = [0.0] * nd, list(np.eye(nd).ravel())
    qform_code, sform_code = (int(v) for v in np.frombuffer(raw, order + "i2", 2, 252))
    A = None
    if sform_code > 0:
        A = np.frombuffer(raw, order + "f4", 12, 280).reshape(3, 4).astype(np.float64)
    elif qform_code > 0:
        b, c, d, qx, qy, qz = (float(v) for v in np.frombuffer(raw, order + "f4", 6, 256))
        a = np.sqrt(max(0.0, 1.0 - b * b - c * c - d * d))
        R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                      [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                      [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
        qfac = -1.0 if float(pixdim[0]) < 0 else 1.0
        S = np.diag([float(pixdim[1]), float(pixdim[2]), float(pixdim[3]) * qfac])
        A = np.concatenate([R @ S, np.array([[qx], [qy], [qz]])], 1)
    if A is not None and nd == 3:
        flip = np.diag([-1.0, -1.0, 1.0])                     # RAS -> LPS
        M = flip @ A[:, :3]
        sp = np.sqrt((M ** 2).sum(0))
        sp[sp == 0] = 1.0
        spacing = [float(v) for v in sp]
        direction = list((M / sp).ravel())
        origin = [float(v) for v in flip @ A[:, 3]]
    return Image(arr, spacing, origin, direction)


def read_metaimage(path: str) -> Image:
    """MetaImage reader (.mha, or .mhd with a raw / zlib-compressed data file next to it)"""
    with open(path, "rb") as f:
        raw = f.read()
    fields, pos = {}, 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if "=" in line:
            k, v = (t.strip() for t in line.split("=", 1))
            fields[k] = v
            if k == "ElementDataFile":
                break
    nd = int(fields.get("NDims", 3))
    sizes = [int(v) for v in fields["DimSize"].split()]
    met = {"MET_CHAR": "i1", "MET_UCHAR": "u1", "MET_SHORT": "i2", "MET_USHORT": "u2", "MET_INT": "i4", "MET_UINT": "u4",
           "MET_LONG_LONG": "i8", "MET_ULONG_LONG": "u8", "MET_FLOAT": "f4", "MET_DOUBLE": "f8"}
    if fields["ElementType"] not in met:
        raise NotImplementedError("MetaImage ElementType %r" % fields["ElementType"])
    if int(fields.get("ElementNumberOfChannels", 1)) != 1:
        raise NotImplementedError("multi-channel MetaImage")
    msb = fields.get("BinaryDataByteOrderMSB", fields.get("ElementByteOrderMSB", "False")).lower() == "true"
    dt = np.dtype(met[fields["ElementType"]])
    if dt.itemsize > 1:
        dt = dt.newbyteorder(">" if msb else "<")
    if fields["ElementDataFile"] == "LOCAL":
        data = raw[pos:]
    else:
        with open(os.path.join(os.path.dirname(os.path.abspath(path)), fields["ElementDataFile"]), "rb") as f:
            data = f.read()
    if fields.get("CompressedData", "False").lower() == "true":
        import zlib
        data = zlib.decompress(data)
    arr = np.frombuffer(data, dt, int(np.prod(sizes))).reshape(sizes[::-1]).astype(dt.newbyteorder("="), copy=True)
    spacing = [float(v) for v in fields.get("ElementSpacing", fields.get("ElementSize", " ".join(["1"] * nd))).split()]
    origin = [float(v) for v in fields.get("Offset", fields.get("Position", fields.get("Origin", " ".join(["0"] * nd)))).split()]
    tm = fields.get("TransformMatrix", fields.get("Orientation", fields.get("Rotation")))
    direction = list(np.eye(nd).ravel())
    if tm is not None:
        direction = list(np.array([float(v) for v in tm.split()]).reshape(nd, nd).T.ravel())   # file holds columns
    return Image(arr, spacing, origin, direction)


def read_image(path: str) -> Image:
    """dispatch on the file name: .nrrd / .nhdr-less NRRD, .nii / .nii.gz, .mha / .mhd"""
    low = path.lower()
    if low.endswith((".nii", ".nii.gz")):
        return read_nifti(path)
    if low.endswith((".mha", ".mhd")):
        return read_metaimage(path)
    return read_nrrd(path)
