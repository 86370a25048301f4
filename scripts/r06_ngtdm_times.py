import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from pyradiomics_amd import engine
dev = torch.device("cuda", 0)
for Ng in (32, 200):
    l, m = bench.make_volume(256, Ng if Ng <= 64 else 64, "uniform", 3, dev)
    if Ng > 64:
        g = torch.Generator(device=dev); g.manual_seed(5)
        l = torch.randint(1, Ng + 1, (256, 256, 256), generator=g, device=dev, dtype=torch.int32)
    for dist in ((1,), (1, 2), (1, 2, 3)):
        for _ in range(2): engine.ngtdm(l, m, Ng, dist)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): engine.ngtdm(l, m, Ng, dist)
        torch.cuda.synchronize()
        print("NGTDM Ng %d dist %s: %.3f ms  path %s" % (Ng, dist, (time.perf_counter() - t0) / 5 * 1e3, engine.last_path()))
