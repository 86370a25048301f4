"""round 4: cProfile of the host thread of a 256^3 case, sorted by cumulative time (functions of this package only)"""
import cProfile, os, pstats, sys, time, io
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
ts = []
for _ in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ex.execute(Image(vol), Image(mask))
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("plain: min %.2f median %.2f ms per case" % (min(ts), sorted(ts)[len(ts) // 2]))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    ex.execute(Image(vol), Image(mask))
torch.cuda.synchronize()
pr.disable()
out = io.StringIO()
st = pstats.Stats(pr, stream=out)
st.sort_stats("cumulative").print_stats(70)
for line in out.getvalue().splitlines():
    if "pyradiomics_amd" in line or "ncalls" in line or "function calls" in line or "{method" in line or "{built-in" in line:
        print(line[:170])
