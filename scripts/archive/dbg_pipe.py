import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_fw import _levels, _mask
from pyradiomics_amd import engine
Ng = 32
shapes = [(24, 30, 512), (24, 30, 512), (20, 26, 256), (18, 22, 300), (24, 30, 512), (30, 30, 64), (24, 30, 512)]
kinds = ["uniform", "smooth", "uniform", "blobs", "flat", "uniform", "smooth"]
masks = ["full", "ball", "random", "full", "full", "random", "ball"]
vols = [(_levels(70 + i, s, Ng, k), _mask(80 + i, s, m)) for i, (s, k, m) in enumerate(zip(shapes, kinds, masks))]
dev = [(torch.from_numpy(i).cuda(), torch.from_numpy(m.astype(np.uint8)).cuda()) for i, m in vols]
want = []
for i, m in dev:
    g, r, ang = engine.glcm_glrlm(i, m, Ng, 512)
    want.append((g.clone(), r.clone()))
def run(idx):
    got = [engine.glcm_glrlm(dev[j][0], dev[j][1], Ng, 512, deferred=True) for j in idx]
    try:
        engine.deferred_status()
    except Exception as e:
        print("status raised", e)
    for j, (g, r, ang) in zip(idx, got):
        dg = (g != want[j][0]); dr = (r != want[j][1])
        if dg.any() or dr.any():
            ag = sorted(set(torch.nonzero(dg)[:, 2].tolist())); ar = sorted(set(torch.nonzero(dr)[:, 2].tolist()))
            print("  vol", j, shapes[j], kinds[j], masks[j], "GLCM diff angles", [tuple(ang[a]) for a in ag], int(dg.sum()),
                  "GLRLM diff angles", [tuple(ang[a]) for a in ar], int(dr.sum()), "sum got/want", float(g.sum()), float(want[j][0].sum()))
        else:
            print("  vol", j, "ok")
engine.set_deferred_mode(1)
for idx in ([0, 1, 2, 3, 4, 5, 6], [0, 0, 0], [1, 1], [2, 2], [3, 3], [4, 4], [0, 1], [1, 0], [6, 6, 6]):
    print("sequence", idx)
    run(idx)
