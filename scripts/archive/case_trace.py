"""where one 256^3 case spends its wall time on the host thread (monkeypatched timers around the phases of
featureextractor.computeFeatures); ms per case, median of 5"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_volume
from pyradiomics_amd import featureextractor as fx, base, cmatrices, imageoperations, filters, engine
from pyradiomics_amd.image import Image
acc = collections.defaultdict(float)
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] += time.perf_counter() - t0
    return w
classes = fx.getFeatureClasses()
for cname, cls in classes.items():
    cls.__init__ = timed("init " + cname, cls.__init__)
    cls.execute = timed("execute " + cname, cls.execute)
    cls.enqueue = timed("enqueue " + cname, cls.enqueue)
cmatrices.segment_sync = timed("segment_sync", cmatrices.segment_sync)
fx.RadiomicsFeatureExtractor._startFeatures = timed("_startFeatures (all)", fx.RadiomicsFeatureExtractor._startFeatures)
fx.RadiomicsFeatureExtractor._finishFeatures = timed("_finishFeatures (all)", fx.RadiomicsFeatureExtractor._finishFeatures)
imageoperations.cropToTumorMask = timed("crop", imageoperations.cropToTumorMask)
engine.swt_level1 = timed("swt", engine.swt_level1)
engine.bin_image = timed("  (bin_image inside init)", engine.bin_image)
engine.firstorder_stats = timed("  (firstorder_stats inside execute)", engine.firstorder_stats)
engine.glszm_compact = timed("  (glszm_compact inside execute)", engine.glszm_compact)
for fn in ("glcm_glrlm", "deferred_join", "glcm_features", "glcm_mcc", "zone_matrix_features", "gldm", "ngtdm", "ngtdm_features",
           "glszm_features", "result_array", "deferred_mark", "deferred_wait", "roi_minmax", "_neigh_common", "_prep",
           "image_enqueue", "image_wait"):
    if hasattr(engine, fn):
        setattr(engine, fn, timed("    engine." + fn, getattr(engine, fn)))
cmatrices._build_angles = timed("    cmatrices._build_angles", cmatrices._build_angles)
engine._build_angles = timed("    engine._build_angles", engine._build_angles)
N = 256
mask = np.zeros((N, N, N), dtype=np.int16)
zz, yy, xx = np.ogrid[:N, :N, :N]
mask[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1
vol = (make_volume(N, 32, "smooth", 0, torch.device("cuda", 0))[0] * 25).cpu().numpy().astype(np.int16)
ex = fx.RadiomicsFeatureExtractor({"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}})
ex.execute(Image(vol), Image(mask)); ex.execute(Image(vol), Image(mask))
acc.clear()
R = 5
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(R):
    ex.execute(Image(vol), Image(mask))
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / R * 1e3
print("total %.2f ms per case" % tot)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("%-42s %6.2f ms" % (k, v / R * 1e3))
