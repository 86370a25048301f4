"""round 4: what the HIP-event brackets of the timed loop cost -- bench.py's headline loop with the library's timing on (every
launch family of every step bracketed by two events) and off (deferred calls record nothing)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyradiomics_amd import engine
dev = torch.device("cuda", 0)
img, msk = bench.make_volume(512, 32, sys.argv[1] if len(sys.argv) > 1 else "uniform", 0, dev)
outs = [[None, None] for _ in range(4)]
def loop(steps, timed):
    n = [0]
    def step():
        o = outs[n[0] % 4]; n[0] += 1
        g, r, _ = engine.glcm_glrlm(img, msk, 32, 512, out_glcm=o[0], out_glrlm=o[1], deferred=True)
        o[0], o[1] = g, r
    for _ in range(3): step()
    engine.deferred_status(); torch.cuda.synchronize()
    if timed: engine.timing_begin()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    engine.deferred_join(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    engine.deferred_status()
    if timed: engine.timing_end()
    return dt / steps * 1e3
for rep in range(3):
    print("events on: %.4f ms/step   events off: %.4f ms/step" % (loop(20, True), loop(20, False)), flush=True)
print("100 steps, events off: %.4f ms/step" % loop(100, False))
def loop_only(steps):
    n = [0]
    def step():
        o = outs[n[0] % 4]; n[0] += 1
        g, r, _ = engine.glcm_glrlm(img, msk, 32, 512, out_glcm=o[0], out_glrlm=o[1], deferred=True)
        o[0], o[1] = g, r
    for _ in range(3): step()
    engine.deferred_status(); torch.cuda.synchronize()
    engine.timing_begin(only="sweep")
    t0 = time.perf_counter()
    for _ in range(steps): step()
    engine.deferred_join(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    engine.deferred_status()
    k = engine.timing_ms("sweep") / engine.timing_count("sweep")
    engine.timing_end()
    return dt / steps * 1e3, k
for rep in range(3):
    print("sweep family only: %.4f ms/step (kernel %.4f ms)" % loop_only(20), flush=True)
