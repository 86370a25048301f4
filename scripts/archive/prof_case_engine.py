"""wall-clock of every engine call inside one device-resident case"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_volume
from pyradiomics_amd import featureextractor as fx, engine
from pyradiomics_amd.image import Image
N = 256
mask = np.zeros((N, N, N), dtype=np.int16)
zz, yy, xx = np.ogrid[:N, :N, :N]
mask[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1
ex = fx.RadiomicsFeatureExtractor({"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}})
vol = (make_volume(N, 32, "smooth", 0, torch.device("cuda", 0))[0] * 25).cpu().numpy().astype(np.int16)
T = collections.defaultdict(list)
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[name].append((time.perf_counter() - t) * 1e3)
        return r
    return w
for name in ("glcm_glrlm", "gldm", "ngtdm", "glszm_compact", "firstorder_stats", "bin_image", "level_counts", "swt_level1", "glcm"):
    setattr(engine, name, timed(name, getattr(engine, name)))
ex.execute(Image(vol), Image(mask))
T.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
ex.execute(Image(vol), Image(mask))
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("total %.1f ms, engine calls %.1f ms" % (tot * 1e3, sum(sum(v) for v in T.values())))
for k, v in sorted(T.items(), key=lambda kv: -sum(kv[1])):
    print("  %-18s n=%2d sum %6.1f ms  each: %s" % (k, len(v), sum(v), " ".join("%.1f" % x for x in v)))
