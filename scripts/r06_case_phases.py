"""round 6: where the HOST thread of one 256^3 case spends its wall time (Original + 8 wavelet sub-bands, six classes), measured with
perf_counter wrappers around a dozen functions (cProfile's per-call overhead distorts a path of 15 000 small calls per case).
Exclusive times: a wrapped function's time minus the wrapped functions it calls.  usage: python scripts/r06_case_phases.py [cases]"""
import os, sys, time, functools, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_volume
from pyradiomics_amd import featureextractor as fx, imageoperations, engine, base, cmatrices, filters
from pyradiomics_amd.image import Image

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
acc, cnt, stack = {}, {}, []


def wrap(owner, name, label=None):
    f = getattr(owner, name)
    label = label or name
    raw = f.__func__ if isinstance(f, (staticmethod, classmethod)) else f

    @functools.wraps(raw)
    def g(*a, **k):
        t0 = time.perf_counter()
        stack.append(0.0)
        try:
            return raw(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            inner = stack.pop()
            acc[label] = acc.get(label, 0.0) + dt - inner
            cnt[label] = cnt.get(label, 0) + 1
            if stack:
                stack[-1] += dt
    setattr(owner, name, staticmethod(g) if isinstance(owner.__dict__.get(name), staticmethod) else g)


E = fx.RadiomicsFeatureExtractor
for owner, name in [(E, "execute"), (E, "loadImage"), (E, "_startFeatures"), (E, "_finishFeatures"), (E, "_queueFeatures"),
                    (imageoperations, "cropToTumorMask"), (imageoperations, "roiTensor"), (imageoperations, "boundingBox"),
                    (engine, "image_enqueue"), (engine, "image_wait"), (engine, "bin_image"), (engine, "swt_level1"),
                    (engine, "firstorder_stats"), (base.RadiomicsFeaturesBase, "__init__"), (base.RadiomicsFeaturesBase, "execute"),
                    (base.RadiomicsFeaturesBase, "_applyBinningDevice"), (base.RadiomicsFeaturesBase, "_calculateSegment"),
                    (base.RadiomicsFeaturesBase, "_fusedSegmentFeatures"), (base.RadiomicsFeaturesBase, "_calculateFeatures")]:
    try:
        wrap(owner, name, "%s.%s" % (getattr(owner, "__name__", "?").split(".")[-1], name))
    except AttributeError as e:
        print("skip", owner, name, e)

N = 256
mask = np.zeros((N, N, N), dtype=np.int16)
zz, yy, xx = np.ogrid[:N, :N, :N]
mask[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1
vol = (make_volume(N, 32, "smooth", 0, torch.device("cuda", 0))[0] * 25).cpu().numpy().astype(np.int16)
ex = E({"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}})
for _ in range(3):
    ex.execute(Image(vol), Image(mask))
acc.clear(); cnt.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(ncases):
    ex.execute(Image(vol), Image(mask))
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / ncases * 1e3
print("%.2f ms per case (wrappers on)" % total)
rows = sorted(acc.items(), key=lambda kv: -kv[1])
for k, v in rows:
    print("  %-48s %7.3f ms per case  (%5.1f calls)" % (k, v / ncases * 1e3, cnt[k] / ncases))
print("  %-48s %7.3f" % ("sum of exclusive times", sum(acc.values()) / ncases * 1e3))
