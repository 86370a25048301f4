"""N > 1 path on CPU: two gloo ranks shard a list of cases exactly like `bench.py --gpus 2` / batch mode shard
volumes over GPUs (no data-path collective), and rank 0 recovers every result in input order."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker_fn(case):
    """one case = (seed, shape): GLCM+GLRLM of a seeded volume through the feature classes on the CPU oracle"""
    sys.path.insert(0, ROOT)
    from oracle import binding
    from pyradiomics_amd import backend, glcm, glrlm
    backend.set(binding.port())
    seed, shape = case
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 200, size=shape).astype(np.int16)
    mask = (rng.random(shape) < 0.8).astype(np.int32)
    out = {}
    for cls in (glcm.RadiomicsGLCM, glrlm.RadiomicsGLRLM):
        out.update({cls.__name__ + "_" + k: float(v) for k, v in cls(img, mask, binWidth=25).execute().items()})
    return out


def _rank_main(rank, world, port, cases, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pyradiomics_amd import batch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = batch.run_batch(cases, _worker_fn)
    own = batch.shard_indices(len(cases), rank, world)
    q.put((rank, own, res))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(oracle_port):
    import torch.multiprocessing as mp
    cases = [(s, (5, 9, 8)) for s in range(5)]
    single = [_worker_fn(c) for c in cases]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, cases, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        rank, own, res = q.get(timeout=240)
        got[rank] = (own, res)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == [0, 2, 4] and got[1][0] == [1, 3]        # disjoint, complete shards
    assert got[1][1] is None                                      # only rank 0 holds the gathered list
    assert got[0][1] == single                                    # same values, input order


def test_slab_split_covers_range_with_halo():
    from pyradiomics_amd.batch import split_slabs
    slabs = split_slabs(512, 8, halo=2)
    assert [s[0] for s in slabs] == list(range(0, 512, 64)) and slabs[-1][1] == 512
    assert slabs[0][2] == 0 and slabs[3][2] == 190 and slabs[3][3] == 258 and slabs[-1][3] == 512
    assert sum(hi - lo for lo, hi, _, _ in split_slabs(10, 3)) == 10


def _voxel_rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from helpers import load_case
    from oracle import binding
    from pyradiomics_amd import backend, batch, firstorder, glcm
    backend.set(binding.port())
    dist.init_process_group("gloo", rank=rank, world_size=world)
    image, mask, _ = load_case("breast1")
    kw = dict(binWidth=25, kernelRadius=1, maskedKernel=True, initValue=np.nan, label=1, voxelBatch=50)
    res = {}
    for cls, feats in ((glcm.RadiomicsGLCM, ["JointEntropy", "Contrast"]), (firstorder.RadiomicsFirstOrder, ["Mean", "Entropy", "10Percentile"])):
        maps = batch.voxel_maps_sharded(cls, image, mask, feats, **kw)
        res[cls.__name__] = None if maps is None else {k: v.array for k, v in maps.items()}
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_voxel_maps_sharded_over_two_ranks(oracle_port):
    """config 4's multi-GPU mode: each rank evaluates a slice of the kernel centres, rank 0 assembles the maps"""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_case
    from pyradiomics_amd import backend, firstorder, glcm
    old = backend._cmatrices
    backend.set(oracle_port)
    try:
        image, mask, _ = load_case("breast1")
        kw = dict(binWidth=25, kernelRadius=1, maskedKernel=True, initValue=np.nan, label=1, voxelBased=True)
        want = {}
        for cls, feats in ((glcm.RadiomicsGLCM, ["JointEntropy", "Contrast"]), (firstorder.RadiomicsFirstOrder, ["Mean", "Entropy", "10Percentile"])):
            fc = cls(image, mask, **kw)
            for f in feats:
                fc.enableFeatureByName(f)
            want[cls.__name__] = {k: v.array for k, v in fc.execute().items()}
    finally:
        backend.set(old)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_voxel_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1] == {"RadiomicsGLCM": None, "RadiomicsFirstOrder": None}
    for cname, maps in want.items():
        assert set(got[0][cname]) == set(maps)
        for k, v in maps.items():
            assert np.array_equal(got[0][cname][k], v, equal_nan=True), (cname, k)
