"""round 5 (second sitting): is the FIRST timed loop of a process slower than the following ones?  The same 32-level deferred loop
(bench.headline_loop, 20 steps, 3 warm-up steps, no instrumented pass) eight times back to back in a fresh process.
usage: python scripts/r05b_coldstart.py [steps] [warmup]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyradiomics_amd import engine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
im, mk = bench.make_volume(512, 32, "uniform", seed=0, device=dev)
torch.cuda.synchronize()
t00 = time.perf_counter()
rows = []
ab = len(sys.argv) > 3        # third argument: A/B of the remainder-workgroup allocation (PRAD_FW_EXTRA_FIRST=1 = as until round 5a)
for i in range(16 if ab else 8):
    if ab:
        if (i // 4) % 2 == 1:
            os.environ["PRAD_FW_EXTRA_FIRST"] = "1"
        else:
            os.environ.pop("PRAD_FW_EXTRA_FIRST", None)
    el, fam, _ = bench.headline_loop(engine, im, mk, 32, 512, steps, warm, torch.cuda.synchronize, [[None, None] for _ in range(4)], families=False)
    rows.append({"loop": i, "t_since_start_ms": round((time.perf_counter() - t00) * 1e3, 1), "ms_per_step": round(el / steps * 1e3, 4),
                 "kernel_ms": round(fam["sweep"], 4), "extra_first": bool(os.environ.get("PRAD_FW_EXTRA_FIRST"))})
print(json.dumps(rows, indent=1))
