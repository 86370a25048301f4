import os, sys, cProfile, pstats, time
sys.path.insert(0, "/root/repo")
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
ex.execute(Image(vol), Image(mask))
pr = cProfile.Profile(); pr.enable()
ex.execute(Image(vol), Image(mask)); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
