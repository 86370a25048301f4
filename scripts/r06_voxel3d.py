"""round 6: the sliding-window GLCM map kernel at 512^3 (5^3 and 5 x 5 windows, JointEntropy), kernel ms of three calls;
a checksum of the map so that variants can be compared.  usage: python scripts/r06_voxel3d.py [size] [features] [kernelRadius]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyradiomics_amd import engine

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
feats = sys.argv[2].split(",") if len(sys.argv) > 2 else ["JointEntropy"]
radius = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
img, msk = bench.make_volume(size, 32, "smooth", 0, dev)
zz, yy, xx = torch.meshgrid(*[torch.arange(size, device=dev, dtype=torch.int32)] * 3, indexing="ij")
vox = torch.stack([zz.reshape(-1), yy.reshape(-1), xx.reshape(-1)])
del zz, yy, xx
for three_d in (True, False):
    kw = dict(kernelRadius=radius, force2D=not three_d, force2Ddimension=0)
    ms = []
    for _ in range(3):
        res = engine.voxel_glcm_features(img, msk, 32, vox, feats, **kw)
        torch.cuda.synchronize()
        ms.append(round(engine.last_kernel_ms("voxel"), 3))
    assert engine.last_variant() == "slide"
    chk = {f: float(res[f].double().nan_to_num().sum()) for f in feats}
    print(json.dumps({"window": ("%d^3" if three_d else "%dx%d") % ((2 * radius + 1,) * (1 if three_d else 2)), "size": size, "kernel_ms": ms, "checksum": chk}), flush=True)
