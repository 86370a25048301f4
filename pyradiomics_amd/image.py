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
        spacing = [float(s) for s in fields["spacings"].split()]
    origin = [0.0] * nd
    if "space origin" in fields:
        o = fields["space origin"]
        origin = [float(x) for x in o[o.index("(") + 1:o.index(")")].split(",")]
    return Image(arr, spacing, origin, direction)


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
