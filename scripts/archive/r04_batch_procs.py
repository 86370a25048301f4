"""round 4: batch mode with worker PROCESSES instead of host threads on one GPU (each process has its own HIP runtime: no
shared runtime locks, no GIL hand-offs) -- P processes x T threads, 36 cases in total, wall from a common start signal.
usage: r04_batch_procs.py P T      (child mode: r04_batch_procs.py child <ncases> <threads> <seed>)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "child":
    import numpy as np, torch
    import bench
    from pyradiomics_amd import batch
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    from pyradiomics_amd.image import Image
    ncases, threads, seed = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dev = torch.device("cuda", 0)
    N = 256
    params = {"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}}
    zz, yy, xx = np.ogrid[:N, :N, :N]
    roi = np.zeros((N, N, N), dtype=np.int16)
    roi[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1
    ex = RadiomicsFeatureExtractor(params)
    vols = [(bench.make_volume(N, 32, "smooth", 1000 * seed + c, dev)[0] * 25).cpu().numpy().astype(np.int16) for c in range(ncases)]
    one = lambda c: ex.execute(Image(vols[c]), Image(roi))
    one(0); one(0)
    if threads > 1:
        batch.warm_threads(lambda: one(0), threads)
    torch.cuda.synchronize()
    print("ready", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    if threads > 1:
        batch.run_batch(list(range(ncases)), one, threads=threads)
    else:
        for c in range(ncases):
            one(c)
    torch.cuda.synchronize()
    print("%d %.6f" % (ncases, time.perf_counter() - t0), flush=True)
    sys.exit(0)
P, T = int(sys.argv[1]), int(sys.argv[2])
total = 36
per = total // P
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(per), str(T), str(i)], stdin=subprocess.PIPE,
                          stdout=subprocess.PIPE, text=True, cwd=ROOT) for i in range(P)]
for p in procs:
    assert p.stdout.readline().strip() == "ready"
t0 = time.perf_counter()
for p in procs:
    p.stdin.write("go\n"); p.stdin.flush()
done = 0
for p in procs:
    n, _ = p.stdout.readline().split()
    done += int(n)
dt = time.perf_counter() - t0
for p in procs:
    p.wait()
print("%d processes x %d threads: %d cases in %.3f s = %.1f cases/s" % (P, T, done, dt, done / dt), flush=True)
