import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyradiomics_amd import engine
mode = sys.argv[1]
outs = {}
rng = np.random.default_rng(1)
for shape, sp in (((256, 256, 256), (1.0, 1.0, 1.0)), ((37, 53, 70), (0.7, 1.3, 2.0)), ((12, 20, 19), (1.0, 1.0, 1.0)), ((5, 4, 9), (1.0, 1.0, 1.0)), ((64, 100, 36), (1, 1, 1))):
    x = torch.from_numpy(rng.integers(0, 800, size=shape).astype(np.int16)).cuda()
    for sigma in (1.0, 3.0):
        o = engine.log_image(x, sp, sigma)
        outs[(shape, sigma)] = o.cpu().numpy()
        if shape[0] == 256:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): engine.log_image(x, sp, sigma)
            torch.cuda.synchronize(); print(mode, "256^3 sigma", sigma, "%.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
np.savez("/tmp/log_%s.npz" % mode, **{"%s_%s" % k: v for k, v in outs.items()})
