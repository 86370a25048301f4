"""round 6: does the fixed-window walk gain from more waves per SIMD?  The same synchronous call (no pack side job: 49 VGPRs)
under launch shapes set through the tuning overrides PRAD_FW_BUDGET_KB (table size -> length slots), PRAD_FW_BLOCKS (workgroups
of the launch) and PRAD_FW_THREADS (their size); results compared bit for bit with the default shape's.
usage: python scripts/r06_occupancy.py [size] [levels]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyradiomics_amd import engine

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
KNOBS = ("PRAD_FW_BUDGET_KB", "PRAD_FW_BLOCKS", "PRAD_FW_THREADS", "PRAD_FW_PER_WAVE", "PRAD_FW_ROWS_THREADS")
CONFIGS = [
    ("default: 256 x 1024, full table", {}),
    ("256 x 1024, 62 KB table", {"PRAD_FW_BUDGET_KB": "62"}),
    ("512 x 512, 62 KB table (2 WG/CU, 4 waves/SIMD)", {"PRAD_FW_BUDGET_KB": "62", "PRAD_FW_BLOCKS": "512", "PRAD_FW_THREADS": "512"}),
    ("512 x 640, 62 KB table (2 WG/CU, 5 waves/SIMD)", {"PRAD_FW_BUDGET_KB": "62", "PRAD_FW_BLOCKS": "512", "PRAD_FW_THREADS": "640"}),
    ("512 x 768, 62 KB table (2 WG/CU, 6 waves/SIMD)", {"PRAD_FW_BUDGET_KB": "62", "PRAD_FW_BLOCKS": "512", "PRAD_FW_THREADS": "768"}),
    ("512 x 768, 62 KB, 4 chunks per wave", {"PRAD_FW_BUDGET_KB": "62", "PRAD_FW_BLOCKS": "512", "PRAD_FW_THREADS": "768", "PRAD_FW_PER_WAVE": "4"}),
    ("512 x 832, 62 KB table (2 WG/CU, 6.5 waves/SIMD)", {"PRAD_FW_BUDGET_KB": "62", "PRAD_FW_BLOCKS": "512", "PRAD_FW_THREADS": "832"}),
    ("768 x 512, 50 KB table (3 WG/CU, 6 waves/SIMD)", {"PRAD_FW_BUDGET_KB": "50", "PRAD_FW_BLOCKS": "768", "PRAD_FW_THREADS": "512"}),
    ("x angle: 12 waves per workgroup", {"PRAD_FW_ROWS_THREADS": "768"}),
    ("x angle: 16 waves per workgroup", {"PRAD_FW_ROWS_THREADS": "1024"}),
    ("288 x 1024, full table (workgroups queue behind the first 256)", {"PRAD_FW_BLOCKS": "288"}),
    ("384 x 1024, full table", {"PRAD_FW_BLOCKS": "384"}),
    ("512 x 1024, full table", {"PRAD_FW_BLOCKS": "512"}),
    ("252 x 1024 (21 per role)", {"PRAD_FW_BLOCKS": "252"}),
    ("default again", {}),
]
out = {}
for dist in ("uniform", "smooth"):
    im, mk = bench.make_volume(size, levels, dist, seed=0, device=dev)
    ref = None
    rows = []
    for name, env in CONFIGS:
        for kname in KNOBS:
            os.environ.pop(kname, None)
        os.environ.update(env)
        for _ in range(3):
            g, r, _a = engine.glcm_glrlm(im, mk, levels, size)
        torch.cuda.synchronize()
        engine.timing_begin()
        k = 10
        for _ in range(k):
            g, r, _a = engine.glcm_glrlm(im, mk, levels, size)
        torch.cuda.synchronize()
        row = {"config": name, "sweep_ms": round(engine.timing_ms("sweep") / k, 4), "rows_ms": round(engine.timing_ms("rows") / k, 4),
               "device_ms": round(engine.timing_ms(None) / k, 4), "variant": engine.last_variant()}
        engine.timing_end()
        if ref is None:
            ref = (g.clone(), r.clone())
        row["bit_exact_vs_default"] = bool(torch.equal(g, ref[0]) and torch.equal(r, ref[1]))
        rows.append(row)
        print(dist, json.dumps(row), flush=True)
    out[dist] = rows
    del im, mk
for kname in KNOBS:
    os.environ.pop(kname, None)
print(json.dumps(out, indent=1))
