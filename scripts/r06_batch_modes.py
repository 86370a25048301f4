"""round 6: modes.batch of bench.py (256^3 cases, Original + 8 wavelet sub-bands, six classes) under the worker layouts the environment
selects: PRAD_BATCH_PROCS worker processes x PRAD_BATCH_THREADS host threads each (PROCS=0: threads of this process).
usage: python scripts/r06_batch_modes.py [cases] [layout ...]   layout = procs:threads, default 0:1 0:2 0:3 1:1 1:2 2:1 4:1"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 36
layouts = sys.argv[2:] or ["0:1", "0:2", "0:3", "1:1", "1:2", "2:1", "4:1"]
dev = torch.device("cuda", 0)
fence = torch.cuda.synchronize
out = []
for lay in layouts:
    if lay.startswith("si="):          # GIL switch interval (seconds) for the layouts that follow
        sys.setswitchinterval(float(lay[3:]))
        print("switch interval", sys.getswitchinterval(), flush=True)
        continue
    p, t = lay.split(":")
    os.environ["PRAD_BATCH_PROCS"], os.environ["PRAD_BATCH_THREADS"] = p, t
    n, dt, nfeat = bench.mode_batch(dev, 0, cases, fence, 1)
    row = {"layout": lay, "how": bench.mode_batch.how, "cases_s": round(n / dt, 1), "one_thread_ms": round(bench.mode_batch.one_thread_ms, 2), "features": nfeat}
    out.append(row)
    print(json.dumps(row), flush=True)
