"""round 4: seeded random stress of the filter kernels this round rebuilt (fused 3-D SWT, one-pass recursive Gaussian,
float64 LoG) against the CPU restatement oracle/filters_oracle.py (pinned by the reference's notebook values):
random shapes (odd sizes, thin volumes), dtypes, spacings, sigma lists.  usage: python scripts/r04_stress_filters.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import filters_oracle as fo
from pyradiomics_amd import filters
from pyradiomics_amd.image import Image

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time()
n = {"wavelet": 0, "log": 0, "log_skipped": 0}
while time.time() - t0 < budget:
    shape = tuple(int(rng.integers(2, 40)) for _ in range(3))
    dt = rng.choice(["int16", "int32", "float32", "float64"])
    x = (rng.standard_normal(shape) * 300 + 200)
    x = x.astype(dt) if dt.startswith("float") else np.round(x).astype(dt)
    if rng.random() < 0.5:
        wl = str(rng.choice(["coif1", "haar", "db2", "sym2"]))
        got = {name: im.array for im, name, _ in filters.getWaveletImage(x, None, wavelet=wl)}
        ap, ret = fo.swt3(x, wl, axes=(2, 1, 0))
        want = {"wavelet-" + k: v for k, v in ret[0].items()}
        want["wavelet-LLL"] = ap
        assert list(got) == list(want), (shape, wl)
        for k in want:
            np.testing.assert_allclose(got[k], want[k], rtol=1e-12, atol=1e-9, err_msg="%s %s %s %s" % (shape, dt, wl, k))
        n["wavelet"] += 1
    else:
        spacing = tuple(float(rng.choice([0.5, 0.78125, 1.0, 1.5, 3.0])) for _ in range(3))
        sigma = float(rng.choice([0.5, 1.0, 2.0, 3.0, 5.0]))
        out = list(filters.getLoGImage(Image(x, spacing), None, sigma=[sigma]))
        if not out:
            n["log_skipped"] += 1
            continue
        want = fo.laplacian_recursive_gaussian(x, spacing, sigma)
        got = out[0][0].array
        assert got.dtype == (np.float64 if dt == "float64" else np.float32), (dt, got.dtype)
        scale = float(np.abs(want).max()) or 1.0
        tol = 2e-6      # (round 5: float internal images for every input type -- ITK's InternalRealType -- also for float64 inputs)
        assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= tol * scale, \
            "LoG %s %s spacing %s sigma %s: %g of %g" % (shape, dt, spacing, sigma, np.abs(got.astype(np.float64) - want).max(), scale)
        n["log"] += 1
print("filter stress ok: %s in %.0f s" % (n, time.time() - t0))
