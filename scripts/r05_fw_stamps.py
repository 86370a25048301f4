"""round 5: per-phase cycle table of sweep_fw_kernel<true,8,false,PACK> at the headline workload.
Run with PRAD_LIB=build_variants/lib_stamps.so (scripts/build_variant.sh stamps "-DPRAD_FW_STAMPS"): every wave keeps an
s_memtime stamp and charges the cycles since the last one to the phase that just ended (kernels_sweepfw.h, FwClock).
Prints a markdown table: mean cycles per wave and phase, share of the wave's life, per-group / per-column-step figures."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_volume, headline_loop          # noqa: E402
from pyradiomics_amd import engine, _lib              # noqa: E402

PH = ["INIT", "GRAB", "CTRL", "ISSUE", "WAIT_ROWS", "WAIT_PACK", "GROUP", "PACK", "SLOW", "FLUSH", "DRAIN", "TOTAL", "REALTIME",
      "NGROUP", "NGROUP_PACK", "NSLOW"]


def main():
    dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
    n, Ng = 512, 32
    dev = torch.device("cuda", 0)
    img, msk = make_volume(n, Ng, dist, 0, dev)
    outs = [[None, None] for _ in range(4)]
    sec, fam, _ = headline_loop(engine, img, msk, Ng, n, 10, 3, torch.cuda.synchronize, outs, families=False)
    lib = _lib.load()
    fn = lib.prad_debug_fw_stamps
    fn.restype = C.c_int
    nw = 4096
    res = {"dist": dist, "kernel_ms_events": fam["sweep"], "ms_per_step": sec / 10 * 1e3}
    for slot, label in ((1, "with the pack side job"), (0, "last volume: no pack")):
        buf = np.zeros((nw, len(PH)), dtype=np.uint64)
        rc = fn(buf.ctypes.data_as(C.c_void_p), slot, nw)
        assert rc == 1, rc       # PRAD_OK
        a = buf.astype(np.float64)
        live = a[:, PH.index("TOTAL")] > 0
        a = a[live]
        tot = a[:, PH.index("TOTAL")]
        rt = a[:, PH.index("REALTIME")]
        ghz = (tot / np.maximum(rt, 1)).mean() * 0.1          # s_memrealtime ticks at 100 MHz
        ng, ngp, nslow = a[:, PH.index("NGROUP")], a[:, PH.index("NGROUP_PACK")], a[:, PH.index("NSLOW")]
        print("\n### launch %s: %d waves, wave life %.0f cycles mean (max %.0f) = %.3f ms at %.2f GHz (s_memtime / s_memrealtime); "
              "events: %.4f ms per launch" % (label, live.sum(), tot.mean(), tot.max(), tot.max() / ghz / 1e6, ghz, fam["sweep"]))
        print("| phase | mean cycles per wave | share of wave life | note |")
        print("|---|---:|---:|---|")
        acc = 0.0
        for k, name in enumerate(PH[:PH.index("TOTAL")]):
            m = a[:, k].mean()
            acc += m
            note = ""
            if name == "GROUP":
                note = "%.0f groups per wave, %.0f cycles per group = %.1f per column-step (64 per group)" % (ng.mean(), a[:, k].sum() / ng.sum(), a[:, k].sum() / ng.sum() / 64)
            if name == "WAIT_ROWS":
                note = "%.0f cycles per group without pack loads in flight" % (a[:, k].sum() / max((ng - ngp).sum(), 1))
            if name == "WAIT_PACK":
                note = "%.0f cycles per group behind pack loads (%.0f such groups per wave)" % (a[:, k].sum() / max(ngp.sum(), 1), ngp.mean())
            if name == "SLOW":
                note = "%.0f single steps per wave, %.0f cycles each" % (nslow.mean(), a[:, k].sum() / max(nslow.sum(), 1))
            if name == "PACK":
                note = "%.0f cycles per packed unit" % (a[:, k].sum() / max(ngp.sum(), 1))
            print("| %s | %.0f | %.3f | %s |" % (name, m, m / tot.mean(), note))
        print("| (sum of phases) | %.0f | %.3f | |" % (acc, acc / tot.mean()))
        res[label] = {name: float(a[:, k].mean()) for k, name in enumerate(PH)}
        res[label]["ghz"] = float(ghz)
        res[label]["wave_life_max"] = float(tot.max())
        # waves by role (12 roles of ~21 workgroups x 16 waves)
        per_role = [tot[i:i + 16 * 21].mean() for i in range(0, len(tot), 16 * 21)]
        print("\nwave life by role (cycles, in launch order): " + " ".join("%.0f" % x for x in per_role))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r05_fw_stamps_%s.json" % dist), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
