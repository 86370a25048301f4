"""round 4: GLDM / NGTDM kernels at 256^3 and 232^3 (uniform + smooth), device ms of the synchronous calls and the one-pass call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_volume
from pyradiomics_amd import engine
dev = torch.device("cuda", 0)
for n in (256, 232):
    for dist in ("uniform", "smooth"):
        img, msk = make_volume(n, 32, dist, 0, dev)
        for name, fn in (("gldm", lambda: engine.gldm(img, msk, 32)), ("ngtdm", lambda: engine.ngtdm(img, msk, 32)),
                         ("both", lambda: engine.gldm_ngtdm(img, msk, 32))):
            try:
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                ms = []
                for _ in range(10):
                    fn()
                    torch.cuda.synchronize()
                    ms.append(engine.last_kernel_ms("neigh"))
                print("%d %s %s: neigh kernel %.4f ms (min %.4f)" % (n, dist, name, sum(ms) / len(ms), min(ms)), flush=True)
            except Exception as e:
                print(n, dist, name, "failed:", type(e).__name__, e, flush=True)
