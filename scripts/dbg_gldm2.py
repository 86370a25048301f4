import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_volume
from pyradiomics_amd import engine, gldm, glcm, ngtdm, cmatrices
from pyradiomics_amd.image import Image
N = 231
dev = torch.device("cuda", 0)
vol = (make_volume(256, 32, "smooth", 0, dev)[0][:N, :N, :N] * 25).contiguous()
z = torch.arange(N, device=dev) - N / 2
mask = ((z[:, None, None] ** 2 + z[None, :, None] ** 2 + z[None, None, :] ** 2) < (0.49 * N) ** 2).to(torch.int16)
im, mk = Image(tensor=vol.to(torch.int16)), Image(tensor=mask)
def T(label, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print("%-40s %.2f ms" % (label, (time.perf_counter() - t) * 1e3)); return r
for rep in range(2):
    fc = T("gldm ctor (binning)", lambda: gldm.RadiomicsGLDM(im, mk, binCount=32))
    fc.enableAllFeatures()
    P = T("  cmatrices.calculate_gldm", lambda: cmatrices.calculate_gldm(fc.imageArray, fc.maskArray, np.array([1]), fc.coefficients["Ng"], 0, False, 0))
    T("  engine.gldm only", lambda: engine.gldm(fc.imageArray, fc.maskArray, fc.coefficients["Ng"]))
    T("  _calculateMatrix", lambda: fc._calculateMatrix())
    T("  execute", lambda: fc.execute())
    f2 = T("ngtdm ctor", lambda: ngtdm.RadiomicsNGTDM(im, mk, binCount=32))
    T("  ngtdm execute", lambda: f2.execute())
print("--- alternating")
lv, mk2, Ng = fc.imageArray, fc.maskArray, fc.coefficients["Ng"]
for rep in range(3):
    T("glcm_glrlm", lambda: engine.glcm_glrlm(lv, mk2, Ng))
    T("  gldm after sweep", lambda: engine.gldm(lv, mk2, Ng))
    T("  gldm again", lambda: engine.gldm(lv, mk2, Ng))
    T("  ngtdm", lambda: engine.ngtdm(lv, mk2, Ng))
    T("glszm_compact", lambda: engine.glszm_compact(lv, mk2, Ng))
    T("  gldm after glszm", lambda: engine.gldm(lv, mk2, Ng))
    T("firstorder", lambda: engine.firstorder_stats(lv, mk2))
    T("  gldm after firstorder", lambda: engine.gldm(lv, mk2, Ng))
