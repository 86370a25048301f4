"""round 5 (second sitting): standalone per-family device times of synchronous calls -- what a step is made of when nothing
overlaps.  512^3 at 32 / 64 levels (uniform, smooth), HIP events on the launch stream (engine.timing_*).
usage: python scripts/r05b_probe.py [size]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyradiomics_amd import engine

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
out = {}
for levels in (32, 64):
    for dist in ("uniform", "smooth"):
        im, mk = bench.make_volume(size, levels, dist, seed=0, device=dev)
        for _ in range(2):
            engine.glcm_glrlm(im, mk, levels, size)
        torch.cuda.synchronize()
        engine.timing_begin()
        k = 5
        t0 = time.perf_counter()
        for _ in range(k):
            engine.glcm_glrlm(im, mk, levels, size)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / k * 1e3
        row = {f: round(engine.timing_ms(f) / k, 4) for f in ("pack", "sweep", "rows", "finalize")}
        row["device"] = round(engine.timing_ms(None) / k, 4)
        row["wall"] = round(wall, 4)
        row["variant"] = engine.last_variant()
        engine.timing_end()
        out["%d_%s" % (levels, dist)] = row
        del im, mk
print(json.dumps(out, indent=1))
