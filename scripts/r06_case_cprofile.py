"""round 6 (from scripts/archive/r04_case_cprofile.py): cProfile of one 256^3 case on the host thread (Original + 8 wavelet sub-bands, six classes) -- where the Python
time of featureextractor.execute goes; prints the 45 functions with the largest own time"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_volume
from pyradiomics_amd import featureextractor as fx
from pyradiomics_amd.image import Image
N = 256
mask = np.zeros((N, N, N), dtype=np.int16)
zz, yy, xx = np.ogrid[:N, :N, :N]
mask[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1
vol = (make_volume(N, 32, "smooth", 0, torch.device("cuda", 0))[0] * 25).cpu().numpy().astype(np.int16)
ex = fx.RadiomicsFeatureExtractor({"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}})
for _ in range(3):
    ex.execute(Image(vol), Image(mask))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    ex.execute(Image(vol), Image(mask))
torch.cuda.synchronize()
print("plain: %.2f ms per case" % ((time.perf_counter() - t0) / 5 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    ex.execute(Image(vol), Image(mask))
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
