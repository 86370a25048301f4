import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from bench import make_volume
from pyradiomics_amd import featureextractor as fx, engine, _lib
from pyradiomics_amd.image import Image
N = 256
mask = np.zeros((N, N, N), dtype=np.int16)
zz, yy, xx = np.ogrid[:N, :N, :N]
mask[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1
ex = fx.RadiomicsFeatureExtractor({"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}})
vol = (make_volume(N, 32, "smooth", 0, torch.device("cuda", 0))[0] * 25).cpu().numpy().astype(np.int16)
lib = _lib.load()
log = []
for fname in ("prad_calculate_glszm_dev", "prad_glszm_sizes", "prad_fill_glszm_compact_dev"):
    f = getattr(lib, fname)
    def mk(f, fname):
        def w(*a):
            torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a); torch.cuda.synchronize()
            log.append((fname, (time.perf_counter() - t) * 1e3, r)); return r
        return w
    setattr(lib, fname, mk(f, fname))
ex.execute(Image(vol), Image(mask)); log.clear()
ex.execute(Image(vol), Image(mask))
for l in log: print("%-32s %8.2f ms rc=%s" % l)
