#!/usr/bin/env python
"""Config 5 through the batch front end: K synthetic N^3 cases on disk (raw NRRD), Original +
8 wavelet sub-bands, all six feature classes, processed by `--jobs` worker processes on the visible GPUs.
Usage: bench_batch.py [N] [K] [jobs,jobs,...]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    import torch
    from bench import make_volume
    from pyradiomics_amd import scripts
    from pyradiomics_amd.image import Image, write_nrrd
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    jobs_list = [int(j) for j in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3]
    tmp = tempfile.mkdtemp(prefix="prad_batch_")
    zz, yy, xx = np.ogrid[:N, :N, :N]
    mask = (((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2).astype(np.int16)
    write_nrrd(os.path.join(tmp, "mask.nrrd"), Image(mask), compress=False)
    cases = []
    for k in range(K):
        vol = (make_volume(N, 32, "smooth", k, torch.device("cuda", 0))[0] * 25).cpu().numpy().astype(np.int16)
        path = os.path.join(tmp, "img%d.nrrd" % k)
        write_nrrd(path, Image(vol), compress=False)
        cases.append((k + 1, {"Image": path, "Mask": os.path.join(tmp, "mask.nrrd")}))
    params = {"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}}
    import json
    pfile = os.path.join(tmp, "params.json")
    json.dump(params, open(pfile, "w"))
    ngpu = torch.cuda.device_count()
    ref = None
    for jobs in jobs_list:
        t = time.perf_counter()
        res = scripts.process_cases(cases, pfile, {}, "segment", jobs)
        dt = time.perf_counter() - t
        nfeat = len(res[0]) - 2
        if ref is None:
            ref = res
        same = all(str(a[k]) == str(b[k]) for a, b in zip(ref, res) for k in a)
        print("%d cases of %d^3 (ROI %d voxels, 9 images, %d features), %d worker(s) on %d GPU(s): %.2f s wall incl. "
              "worker start-up, %.2f cases/s; rows identical to the first run: %s"
              % (K, N, int(mask.sum()), nfeat, jobs, ngpu, dt, K / dt, same), flush=True)


if __name__ == "__main__":
    main()
