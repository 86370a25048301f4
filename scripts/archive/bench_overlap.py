#!/usr/bin/env python
"""Do two independent GLCM+GLRLM pipelines (two host threads = two library contexts, two HIP streams) overlap on one
GPU?  The pack kernel is HBM-bound, the sweep kernel issue/LDS-bound.  Prints ms per volume with 1 and 2 pipelines."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_volume
from pyradiomics_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = 20
dev = torch.device("cuda", 0)
vols = [make_volume(n, 32, "uniform", s, dev) for s in range(2)]
outs = [(torch.empty((32, 32, 13), dtype=torch.float64, device=dev), torch.empty((32, n, 13), dtype=torch.float64, device=dev)) for _ in range(2)]


def worker(i, steps, stream, barrier):
    with torch.cuda.stream(stream):
        img, m = vols[i]
        for _ in range(3):
            engine.glcm_glrlm(img, m, 32, n, out_glcm=outs[i][0], out_glrlm=outs[i][1], deferred=True)
        stream.synchronize()
        barrier.wait()
        for _ in range(steps):
            engine.glcm_glrlm(img, m, 32, n, out_glcm=outs[i][0], out_glrlm=outs[i][1], deferred=True)
        stream.synchronize()
        engine.deferred_status()


for nthreads in (1, 2, 1, 2):
    streams = [torch.cuda.Stream() for _ in range(nthreads)]
    bar = threading.Barrier(nthreads + 1)
    th = [threading.Thread(target=worker, args=(i, K, streams[i], bar)) for i in range(nthreads)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print("%d pipeline(s): %.3f ms per volume (%.1f Gvox/s)" % (nthreads, dt * 1e3 / (K * nthreads), n ** 3 * K * nthreads / dt / 1e9), flush=True)
