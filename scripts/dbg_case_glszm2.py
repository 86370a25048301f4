import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from bench import make_volume
from pyradiomics_amd import featureextractor as fx, engine, _lib
from pyradiomics_amd.engine import _neigh_common, _iptr, _stream_ptr
from pyradiomics_amd.image import Image
N = 256
mask = np.zeros((N, N, N), dtype=np.int16)
zz, yy, xx = np.ogrid[:N, :N, :N]
mask[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1
ex = fx.RadiomicsFeatureExtractor({"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}})
vol = (make_volume(N, 32, "smooth", 0, torch.device("cuda", 0))[0] * 25).cpu().numpy().astype(np.int16)
log = []
def glszm_compact(image, mask, Ng, Ns=None, force2D=False, force2Ddimension=0):
    marks = [("start", time.perf_counter())]
    def mark(n):
        torch.cuda.synchronize(); marks.append((n, time.perf_counter()))
    torch.cuda.synchronize(); marks = [("start", time.perf_counter())]
    lib, image, mask, size, f2d, angles = _neigh_common(image, mask, None, force2D, force2Ddimension); mark("prep")
    Na, Nd = angles.shape
    nz = C.c_longlong(0)
    rc = lib.prad_calculate_glszm_dev(C.c_void_p(image.data_ptr()), C.c_void_p(mask.data_ptr()), _iptr(size), Nd, _iptr(angles), Na, int(Ng), int(Ns), 1, None, 0, f2d, C.byref(nz), _stream_ptr()); mark("zones")
    maxRegion = max(int(rc), 1)
    cap = int(min(maxRegion, int(np.sqrt(2.0 * image.numel())) + 2))
    sizes = np.empty(cap, dtype=np.intc); mark("np.empty")
    k = lib.prad_glszm_sizes(_iptr(sizes), cap); mark("sizes")
    out = torch.empty((Ng, max(k, 1)), dtype=torch.float64, device=image.device); mark("torch.empty")
    rc = lib.prad_fill_glszm_compact_dev(C.c_void_p(out.data_ptr()), int(Ng), int(k), _stream_ptr()); mark("fill")
    r = out[:, :k], sizes[:k].copy(); mark("slice")
    log.append(" ".join("%s %.2f" % (n, (t - marks[i][1]) * 1e3) for i, (n, t) in enumerate(marks[1:])))
    return r
engine.glszm_compact = glszm_compact
ex.execute(Image(vol), Image(mask)); log.clear()
t = time.perf_counter(); ex.execute(Image(vol), Image(mask)); print("total %.1f" % ((time.perf_counter() - t) * 1e3))
print("\n".join(log))
