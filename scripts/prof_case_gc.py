import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_volume
from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
from pyradiomics_amd.image import Image
N = 256
mask = np.zeros((N, N, N), dtype=np.int16)
zz, yy, xx = np.ogrid[:N, :N, :N]
mask[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1
ex = RadiomicsFeatureExtractor({"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}})
vol = (make_volume(N, 32, "smooth", 0, torch.device("cuda", 0))[0] * 25).cpu().numpy().astype(np.int16)
def run(n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter()
        ex.execute(Image(vol), Image(mask)); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return ["%.1f" % (x * 1e3) for x in ts]
print("default gc      ", run(), gc.get_count(), gc.get_threshold())
gc.freeze()
print("after gc.freeze ", run())
gc.disable()
print("gc disabled     ", run())
