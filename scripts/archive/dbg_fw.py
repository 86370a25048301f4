"""debug helper: per-angle mismatch counts of the fixed-window kernel vs the CPU checker"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyradiomics_amd import cmatrices as cm
from oracle import binding
chk = binding.ref() if binding.have_ref() else binding.port()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_fw import _levels, _mask
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1].split(",")]
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
mkind = sys.argv[3] if len(sys.argv) > 3 else "full"
Ng = int(sys.argv[4]) if len(sys.argv) > 4 else 32
for shape in shapes:
    img, mask = _levels(5, shape, Ng, kind), _mask(2, shape, mkind)
    Nr = max(shape)
    g, r, ang = cm.calculate_glcm_glrlm(img, mask, Ng, Nr, False, 0)
    eg, _ = chk.calculate_glcm(img, mask, [1], Ng, False, 0)
    er, _ = chk.calculate_glrlm(img, mask, Ng, Nr, False, 0)
    print(shape, kind, mkind, "CL", os.environ.get("PRAD_FW_CL"))
    for a in range(ang.shape[0]):
        dg = g[0, :, :, a] - eg[0, :, :, a]
        dr = r[0, :, :, a] - er[0, :, :, a]
        if np.any(dg) or np.any(dr):
            print("  angle", ang[a], "glcm: n=%d sum=%d absum=%d  glrlm: n=%d sum=%d lens=%s" % (
                np.count_nonzero(dg), dg.sum(), np.abs(dg).sum(), np.count_nonzero(dr), dr.sum(),
                sorted(set(np.argwhere(dr != 0)[:, 1].tolist()))[:12]))
