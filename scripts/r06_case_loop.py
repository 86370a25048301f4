"""round 6: N 256^3 cases (Original + 8 wavelet sub-bands, six classes) on ONE host thread -- execute() case after case, or
executeMany (one case of overlap) with MANY=1 -- for the kernel trace (scripts/r06_case_trace.sh).  usage: python scripts/r06_case_loop.py [cases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_volume
from pyradiomics_amd import featureextractor as fx
from pyradiomics_amd.image import Image
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = 256
mask = np.zeros((N, N, N), dtype=np.int16)
zz, yy, xx = np.ogrid[:N, :N, :N]
mask[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1
vol = (make_volume(N, 32, "smooth", 0, torch.device("cuda", 0))[0] * 25).cpu().numpy().astype(np.int16)
ex = fx.RadiomicsFeatureExtractor({"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}})
for _ in range(3):
    ex.execute(Image(vol), Image(mask))
torch.cuda.synchronize(); t0 = time.perf_counter()
if os.environ.get("MANY"):
    res = list(ex.executeMany((Image(vol), Image(mask)) for _ in range(n)))
else:
    res = [ex.execute(Image(vol), Image(mask)) for _ in range(n)]
torch.cuda.synchronize()
print("%s: %.2f ms per case" % ("executeMany" if os.environ.get("MANY") else "execute", (time.perf_counter() - t0) / n * 1e3))
