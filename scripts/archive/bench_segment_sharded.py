#!/usr/bin/env python
"""One large segment over several GPUs (third row; pyradiomics_amd.batch.segment_matrices_sharded).
Under a torch.distributed launch (RCCL) every rank times the real thing.  On ONE GPU (no launch) the script plays
ranks 0..world-1 in turn and reports every share's device time: max over ranks = the compute time of the split
(the exchange step -- all-reduce of ~2 MB -- is not in it), sum over ranks / single-GPU time = the split's overhead."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_volume
from pyradiomics_amd import batch, engine

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--levels", type=int, default=32)
ap.add_argument("--dist", default="uniform")
ap.add_argument("--world", type=int, default=8, help="ranks to play when not launched under torch.distributed")
ap.add_argument("--classes", default="glcm,glrlm,gldm,ngtdm,glszm")
a = ap.parse_args()
classes = tuple(a.classes.split(","))
launched = int(os.environ.get("WORLD_SIZE", 1)) > 1
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best * 1e3


if launched:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=dev)
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == 0:
        img, msk = make_volume(a.size, a.levels, a.dist, 0, dev)
        t = time.perf_counter()
        img, msk = batch.replicate_volume(img, msk, 0)
    else:
        t = time.perf_counter()
        img, msk = batch.replicate_volume(None, None, 0, device=dev)
    torch.cuda.synchronize()
    t_rep = (time.perf_counter() - t) * 1e3
    for cls in classes:
        ms = timed(lambda: batch.segment_matrices_sharded(img, msk, a.levels, classes=(cls,)))
        if rank == 0:
            print("%d ranks, %d^3 %s: %-6s %.3f ms (split + exchange)" % (world, a.size, a.dist, cls, ms), flush=True)
    if rank == 0:
        print("replicate_volume (broadcast of %d MB): %.2f ms" % (img.numel() >> 20, t_rep))
    dist.destroy_process_group()
else:
    img, msk = make_volume(a.size, a.levels, a.dist, 0, dev)
    single = {
        "glcm": lambda: engine.glcm_glrlm(img, msk, a.levels, want_glrlm=False),
        "glrlm": lambda: engine.glcm_glrlm(img, msk, a.levels, want_glcm=False),
        "gldm": lambda: engine.gldm(img, msk, a.levels),
        "ngtdm": lambda: engine.ngtdm(img, msk, a.levels),
        "glszm": lambda: engine.glszm_compact(img, msk, a.levels),
    }
    both = [c for c in ("glcm", "glrlm") if c in classes]
    groups = ([tuple(both)] if len(both) == 2 else [(c,) for c in both]) + [(c,) for c in classes if c not in both]
    for grp in groups:
        if grp == ("glcm", "glrlm"):
            t1 = timed(lambda: engine.glcm_glrlm(img, msk, a.levels))
        else:
            t1 = timed(single[grp[0]])
        shares = [timed(lambda r=r: batch.segment_partials(img, msk, a.levels, r, a.world, classes=grp))
                  for r in range(a.world)]
        print("%d^3 %s %-10s single GPU %.3f ms | %d shares: max %.3f ms, sum %.3f ms -> compute speed-up %.2fx"
              % (a.size, a.dist, "+".join(grp), t1, a.world, max(shares), sum(shares), t1 / max(shares)), flush=True)
