"""round 4: LoG of five sigmas at 256^3 -- all five in the same launches vs groups of 3 + 2, 2 + 2 + 1, one by one (does a
smaller in-flight working set let the second sweep of a pass hit the Infinity Cache?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_volume
from pyradiomics_amd import engine
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lv, msk = make_volume(size, 32, "smooth", 0, torch.device("cuda", 0))
img = (lv.to(torch.float32) * 25.0 + 3.0).to(torch.int16)
sig = (1.0, 2.0, 3.0, 4.0, 5.0)
for groups in ((5,), (3, 2), (2, 2, 1), (1, 1, 1, 1, 1)):
    def run():
        i = 0
        for g in groups:
            engine.log_images(img, (1.0, 1.0, 1.0), sig[i:i + g]); i += g
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize()
    print("%d^3 groups %s: %.3f ms per five sigmas" % (size, groups, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
