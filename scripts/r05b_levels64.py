"""round 5 (second sitting): the deferred loop of bench.py's modes.levels64 -- the two-table walk with the NEXT volume's pack
as a side job (pipeline, default) against the lanes route of rounds 3-5a (PRAD_FW2_LANES=1), and the 32-level step beside them.
usage: python scripts/r05b_levels64.py [size] [steps] [ENV=VAL,ENV=VAL ...]     (one extra 64-level run per environment set)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyradiomics_amd import engine

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
extra = sys.argv[3:]
dev = torch.device("cuda:0")


def fence():
    torch.cuda.synchronize()


def run(levels, dist, env):
    for kv in env:
        k, v = kv.split("=")
        os.environ[k] = v
    im, mk = bench.make_volume(size, levels, dist, seed=0, device=dev)
    el, fam, _ = bench.headline_loop(engine, im, mk, levels, size, steps, 3, fence, [[None, None] for _ in range(4)])
    for kv in env:
        os.environ.pop(kv.split("=")[0], None)
    del im, mk
    return {"ms_per_step": round(el / steps * 1e3, 4), "variant": engine.last_variant(), **{k: round(v, 4) for k, v in fam.items()}}


out = {"lib": os.environ.get("PRAD_LIB", "default")}
out["32_uniform"] = run(32, "uniform", [])
for dist in ("uniform", "smooth"):
    out["64_%s_pipeline" % dist] = run(64, dist, [])
    out["64_%s_lanes" % dist] = run(64, dist, ["PRAD_FW2_LANES=1"])
for e in extra:
    lv = 64
    if e.startswith("32:"):          # "32:ENV=VAL": the same at 32 levels
        lv, e = 32, e[3:]
    env = e.split(",")
    for dist in ("uniform", "smooth"):
        key = "%d_%s_%s" % (lv, dist, e)
        while key in out:
            key += "_again"
        out[key] = run(lv, dist, env)
out["64_uniform_pipeline_again"] = run(64, "uniform", [])
print(json.dumps(out, indent=1))
