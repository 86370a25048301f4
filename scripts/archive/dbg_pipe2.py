import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_fw import _levels, _mask
from pyradiomics_amd import engine
from oracle import binding
chk = binding.ref() if binding.have_ref() else binding.port()
Ng = 32
shapes = [(24, 30, 512), (24, 30, 512), (20, 26, 256), (18, 22, 300), (24, 30, 512), (30, 30, 64), (24, 30, 512)]
kinds = ["uniform", "smooth", "uniform", "blobs", "flat", "uniform", "smooth"]
masks = ["full", "ball", "random", "full", "full", "random", "ball"]
vols = [(_levels(70 + i, s, Ng, k), _mask(80 + i, s, m)) for i, (s, k, m) in enumerate(zip(shapes, kinds, masks))]
dev = [(torch.from_numpy(i).cuda(), torch.from_numpy(m.astype(np.uint8)).cuda()) for i, m in vols]
def report(tag, j, g, r, ang):
    img, mask = vols[j]
    eg = chk.calculate_glcm(img, mask, [1], Ng, False, 0)[0][0]
    er = chk.calculate_glrlm(img, mask, Ng, 512, False, 0)[0][0]
    g = g.cpu().numpy(); r = r.cpu().numpy()
    dg = g != eg; dr = r != er
    if dg.any() or dr.any():
        ag = sorted(set(np.argwhere(dg)[:, 2].tolist())); ar = sorted(set(np.argwhere(dr)[:, 2].tolist()))
        print(" ", tag, "vol", j, "GLCM diff angles", [tuple(ang[a]) for a in ag], int(dg.sum()), "GLRLM diff angles", [tuple(ang[a]) for a in ar], int(dr.sum()))
        for a in ag[:3]:
            idx = np.argwhere(dg[:, :, a])[:4]
            print("     angle", tuple(ang[a]), [(tuple(i), g[i[0], i[1], a], eg[i[0], i[1], a]) for i in idx])
    else:
        print(" ", tag, "vol", j, "ok")
mode = sys.argv[1] if len(sys.argv) > 1 else "timing"
engine.set_deferred_mode(1)
if mode == "timing":
    engine.timing_begin()
got = [engine.glcm_glrlm(i, m, Ng, 512, deferred=True) for i, m in dev]
engine.deferred_status()
for j, (g, r, ang) in enumerate(got):
    report("pipe", j, g, r, ang)
for j, (i, m) in enumerate(dev):
    g, r, ang = engine.glcm_glrlm(i, m, Ng, 512)
    report("sync", j, g, r, ang)
