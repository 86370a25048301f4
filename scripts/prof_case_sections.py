"""wall-clock split of one device-resident case (Original + 8 wavelet sub-bands, six classes) by section"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_volume
from pyradiomics_amd import featureextractor as fx, imageoperations, filters, base
from pyradiomics_amd.image import Image
N = 256
mask = np.zeros((N, N, N), dtype=np.int16)
zz, yy, xx = np.ogrid[:N, :N, :N]
mask[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1
ex = fx.RadiomicsFeatureExtractor({"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}})
vol = (make_volume(N, 32, "smooth", 0, torch.device("cuda", 0))[0] * 25).cpu().numpy().astype(np.int16)
T = collections.defaultdict(float)
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[name] += time.perf_counter() - t
        return r
    return w
classes = fx.getFeatureClasses()
for cname, cls in classes.items():
    cls.__init__ = timed(cname + ".init(bin)", cls.__init__)
    cls._initCalculation = timed(cname + ".matrix+coef", cls._initCalculation)
    cls.execute = timed(cname + ".execute(total)", cls.execute)
imageoperations.cropToTumorMask = timed("crop", imageoperations.cropToTumorMask)
ex.execute(Image(vol), Image(mask))
T.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
ex.execute(Image(vol), Image(mask))
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("total %.1f ms" % (tot * 1e3))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-28s %7.1f ms" % (k, v * 1e3))
