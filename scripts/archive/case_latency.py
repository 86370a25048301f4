"""single-thread latency of one 256^3 case (Original + 8 wavelet sub-bands, six classes), with and without the case
pipeline (enqueueSegment); run under rocprofv3 --kernel-trace --stats for the GPU-busy share"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_volume
from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
from pyradiomics_amd.image import Image
N = int(os.environ.get("CASE_N", "256"))
REP = int(os.environ.get("CASE_REP", "6"))
mask = np.zeros((N, N, N), dtype=np.int16)
zz, yy, xx = np.ogrid[:N, :N, :N]
mask[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1
vol = (make_volume(N, 32, "smooth", 0, torch.device("cuda", 0))[0] * 25).cpu().numpy().astype(np.int16)
for on in ([True, False] if not os.environ.get("CASE_ONLY") else [os.environ["CASE_ONLY"] == "1"]):
    ex = RadiomicsFeatureExtractor({"setting": {"binCount": 32, "additionalInfo": False, "enqueueSegment": on},
                                    "imageType": {"Original": {}, "Wavelet": {}}})
    ex.execute(Image(vol), Image(mask))
    ts = []
    for _ in range(REP):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = ex.execute(Image(vol), Image(mask))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("enqueueSegment=%s: %d features, ms per case: min %.2f median %.2f" % (on, len(r), min(ts), sorted(ts)[len(ts) // 2]), flush=True)
