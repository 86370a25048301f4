"""GLDM / NGTDM matrix kernels on a 232^3 smooth volume with a ball ROI (the case pipeline's shape): device ms per call"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_volume
from pyradiomics_amd import engine
N = int(os.environ.get("NEIGH_N", "232"))
dev = torch.device("cuda", 0)
for dist in ("smooth", "uniform"):
    lev = make_volume(N, 32, dist, 0, dev)[0].to(torch.int32)
    zz, yy, xx = np.ogrid[:N, :N, :N]
    m = (((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.49 * N) ** 2)
    msk = torch.from_numpy(m.astype(np.uint8)).to(dev)
    for name, fn in (("gldm", lambda: engine.gldm(lev, msk, 32)), ("ngtdm", lambda: engine.ngtdm(lev, msk, 32))):
        fn()
        ts = []
        for _ in range(5):
            fn()
            ts.append(engine.last_kernel_ms("neigh"))
        print("%s %s %d^3: neigh kernel %.1f us" % (name, dist, N, 1e3 * min(ts)), flush=True)
