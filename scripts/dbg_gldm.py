import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyradiomics_amd import engine
dev = torch.device("cuda", 0)
for N, ball in ((256, False), (256, True), (231, False), (231, True), (232, True)):
    g = torch.Generator(device=dev); g.manual_seed(0)
    img = torch.randint(1, 33, (N, N, N), generator=g, device=dev, dtype=torch.int32)
    if ball:
        z = torch.arange(N, device=dev) - N / 2
        mask = ((z[:, None, None] ** 2 + z[None, :, None] ** 2 + z[None, None, :] ** 2) < (0.49 * N) ** 2)
    else:
        mask = torch.ones((N, N, N), dtype=torch.bool, device=dev)
    for name, fn in (("gldm", lambda: engine.gldm(img, mask, 32)), ("ngtdm", lambda: engine.ngtdm(img, mask, 32))):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize()
        print(N, "ball" if ball else "full", name, "%.2f ms" % ((time.perf_counter() - t) / 3 * 1e3), engine.last_path(), "kernel %.2f" % engine.last_kernel_ms(None))
