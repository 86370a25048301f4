"""round 5: second mismatch of the walk (no-pad rows): variations of the 256-wide crop of the regression fixture"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding
from pyradiomics_amd import cmatrices as cm, _lib

ck = binding.ref() if binding.have_ref() else binding.port()
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(os.path.join(R, "tests", "golden", "regress", "fw_long_runs_138x58x300.npz"))
img0 = np.ascontiguousarray(d["img"].astype(np.int32)[:, :, :256])
Ng = int(d["Ng"])


def run(tag, img, env=None):
    env = env or {}
    mask = np.ones(img.shape, bool)
    for k, v in env.items():
        os.environ[k] = v
    try:
        Nr = max(img.shape)
        g, r, ang = cm.calculate_glcm_glrlm(img, mask, Ng, Nr, False, 0)
    finally:
        for k in env:
            del os.environ[k]
    wr, _ = ck.calculate_glrlm(img, mask, Ng, Nr, False, 0)
    dr = (r - wr).reshape(r.shape[-3:])
    cells = np.argwhere(dr != 0)
    print("%-26s shape %s: %s" % (tag, img.shape, "OK" if len(cells) == 0 else "MISMATCH (level-1, len-1, angle, got-want): %s" % [
        (int(i), int(j), int(a), int(dr[i, j, a])) for i, j, a in cells[:24]]), flush=True)


run("as is", img0)
run("one piece", img0, {"PRAD_FW_CL": "144"})
run("CL 32", img0, {"PRAD_FW_CL": "32"})
run("CL 16", img0, {"PRAD_FW_CL": "16"})
run("CL 8", img0, {"PRAD_FW_CL": "8"})
run("RS 16", img0, {"PRAD_FW_RS": "16"})
run("RS 30", img0, {"PRAD_FW_RS": "30"})
for z1 in (128, 100, 80, 72, 64):
    run("z[:%d]" % z1, np.ascontiguousarray(img0[:z1]))
run("z[20:138]", np.ascontiguousarray(img0[20:]))
run("z[64:138]", np.ascontiguousarray(img0[64:]))
run("y[:44]", np.ascontiguousarray(img0[:, :44]))
run("y[14:]", np.ascontiguousarray(img0[:, 14:]))
run("x[:128]", np.ascontiguousarray(img0[:, :, :128]))
run("x[128:256]", np.ascontiguousarray(img0[:, :, 128:256]))
