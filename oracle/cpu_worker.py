"""TEST / BENCH INFRASTRUCTURE ONLY -- one CPU worker of the all-cores baseline in bench.py: builds its own synthetic
slab (uniform levels 1..Ng, full mask, the bench volume's distribution), times the reference's own C
(oracle/_ref, or the restatement if that prebuilt file is absent) on calculate_glcm + calculate_glrlm, single
threaded like the reference, and prints `voxels seconds`.  The reference parallelises the same way: one process per
case (radiomics/scripts/__init__.py:387-416).  Usage: python -m oracle.cpu_worker <nz> <ny> <nx> <Ng> <seed>"""
import sys
import time

import numpy as np


def main():
    nz, ny, nx, Ng, seed = (int(a) for a in sys.argv[1:6])
    from oracle import binding
    cpu = binding.ref() if binding.have_ref() else binding.port()
    rng = np.random.default_rng(seed)
    img = rng.integers(1, Ng + 1, size=(nz, ny, nx), dtype=np.int32)
    msk = np.ones(img.shape, dtype=bool)
    print("ready", flush=True)
    sys.stdin.readline()                       # all workers start together
    t = time.perf_counter()
    cpu.calculate_glcm(img, msk, [1], Ng, False, 0)
    cpu.calculate_glrlm(img, msk, Ng, int(max(img.shape)), False, 0)
    print("%d %.6f" % (img.size, time.perf_counter() - t), flush=True)


if __name__ == "__main__":
    main()
