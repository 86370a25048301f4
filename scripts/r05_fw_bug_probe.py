"""round 5: localise the walk-kernel mismatch the stress found (gpurun_out/r05_stress_fail_1.npz): variations of the case"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding
from pyradiomics_amd import cmatrices as cm, _lib

ck = binding.ref() if binding.have_ref() else binding.port()
d = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "regress", "fw_long_runs_138x58x300.npz"))
img0, mask0, Ng = d["img"].astype(np.int32), d["mask"], int(d["Ng"])


def run(tag, img, mask, env=None):
    env = env or {}
    for k, v in env.items():
        os.environ[k] = v
    try:
        Nr = max(img.shape)
        g, r, ang = cm.calculate_glcm_glrlm(img, mask, Ng, Nr, False, 0)
        var = _lib.last_variant()
    finally:
        for k in env:
            del os.environ[k]
    wg, _ = ck.calculate_glcm(img, mask, [1], Ng, False, 0)
    wr, _ = ck.calculate_glrlm(img, mask, Ng, Nr, False, 0)
    bad = sorted(set(np.argwhere(g != wg)[:, -1].tolist()) | set(np.argwhere(r != wr)[:, -1].tolist()))
    msg = "%-28s variant %-5s shape %s: %s" % (tag, var, img.shape, "OK" if not bad else "MISMATCH angles %s" % bad)
    if bad:
        a = bad[0]
        dr = (r - wr)[0, :, :, a] if r.ndim == 4 else (r - wr)[:, :, a]
        cells = np.argwhere(dr != 0)
        msg += " |dG| %g |dR| %g; GLRLM cells (level-1, len-1, got-want): %s" % (
            np.abs(g - wg).sum(), np.abs(r - wr).sum(), [(int(i), int(j), int(dr[i, j])) for i, j in cells[:14]])
    print(msg, flush=True)


for rep in range(3):
    run("as is #%d" % rep, img0, mask0)
run("PRAD_NO_FW", img0, mask0, {"PRAD_NO_FW": "1"})
run("one piece (CL 144)", img0, mask0, {"PRAD_FW_CL": "144"})
run("CL 32", img0, mask0, {"PRAD_FW_CL": "32"})
run("CL 16", img0, mask0, {"PRAD_FW_CL": "16"})
run("RS 16", img0, mask0, {"PRAD_FW_RS": "16"})
run("x cropped to 256", np.ascontiguousarray(img0[:, :, :256]), np.ascontiguousarray(mask0[:, :, :256]))
run("x cropped to 296", np.ascontiguousarray(img0[:, :, :296]), np.ascontiguousarray(mask0[:, :, :296]))
big = np.concatenate([img0, img0[:, :, :212]], axis=2)
run("x tiled to 512", big, np.ones(big.shape, bool))
run("z cropped to 128", np.ascontiguousarray(img0[:128]), np.ascontiguousarray(mask0[:128]))
run("z cropped to 64", np.ascontiguousarray(img0[:64]), np.ascontiguousarray(mask0[:64]))
run("y cropped to 8", np.ascontiguousarray(img0[:, :8]), np.ascontiguousarray(mask0[:, :8]))
const = np.full_like(img0, 5)
run("constant volume", const, mask0)
run("constant, one piece", const, mask0, {"PRAD_FW_CL": "144"})
