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


# ---- one segment over all ranks: angle / z-slab / level split + one exchange step -----------------------------
def _segment_inputs():
    rng = np.random.default_rng(11)
    shape = (9, 12, 10)
    field = rng.random(shape)
    for ax in range(3):                              # a little smoothing: zones and runs longer than one voxel
        field = field + np.roll(field, 1, axis=ax)
    image = (1 + np.floor((field - field.min()) / (np.ptp(field) + 1e-9) * 6)).astype(np.int32)   # levels 1..6
    mask = (rng.random(shape) < 0.85).astype(np.uint8)
    return image, mask, 6


def _segment_rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import binding
    from oracle.segment_ops import OracleSegmentOps
    from pyradiomics_amd import batch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    image, mask, Ng = _segment_inputs()
    if rank == 0:                                    # only rank 0 holds the volume: one broadcast replicates it
        img_t, msk_t = batch.replicate_volume(torch.from_numpy(image), torch.from_numpy(mask), src=0)
    else:
        img_t, msk_t = batch.replicate_volume(None, None, src=0, device="cpu")
    res = batch.segment_matrices_sharded(img_t, msk_t, Ng, alpha=1, ops=OracleSegmentOps(binding.port()))
    part = batch.segment_partials(img_t, msk_t, Ng, rank, world, alpha=1, ops=OracleSegmentOps(binding.port()))
    q.put((rank, {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in res.items()},
           {k: v.numpy() for k, v in part.items() if k in ("glcm", "gldm_acc")}))
    dist.barrier()
    dist.destroy_process_group()


def test_one_segment_over_two_ranks_matches_single_process(oracle_port):
    import torch.multiprocessing as mp
    image, mask, Ng = _segment_inputs()
    cm = oracle_port
    glcm, angles = cm.calculate_glcm(image, mask, [1], Ng, False, 0)
    glrlm, _ = cm.calculate_glrlm(image, mask, Ng, max(image.shape), False, 0)
    gldm = cm.calculate_gldm(image, mask, [1], Ng, 1, False, 0)
    ngtdm = cm.calculate_ngtdm(image, mask, [1], Ng, False, 0)
    glszm = cm.calculate_glszm(image, mask, Ng, int(mask.sum()), False, 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 977) % 2000
    procs = [ctx.Process(target=_segment_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        rank, res, part = q.get(timeout=240)
        got[rank] = (res, part)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):                              # every rank ends with the whole result
        res = got[rank][0]
        assert np.array_equal(res["glcm_angles"], angles)
        assert np.array_equal(res["glcm"], glcm[0])
        assert np.array_equal(res["glrlm"], glrlm[0])
        assert np.array_equal(res["gldm"], gldm[0])
        assert np.array_equal(res["ngtdm"][:, [0, 2]], ngtdm[0][:, [0, 2]])
        np.testing.assert_allclose(res["ngtdm"][:, 1], ngtdm[0][:, 1], rtol=1e-12)   # raster-order float sum
        P, sizes = res["glszm"]
        dense = np.zeros_like(glszm[0])
        dense[:, sizes - 1] = P
        assert np.array_equal(dense, glszm[0])
    # the shares really are disjoint: angle columns of the pair counts, planes of the dependence accumulators
    a0, a1 = got[0][1]["glcm"], got[1][1]["glcm"]
    assert not a0[:, :, 1::2].any() and not a1[:, :, 0::2].any() and a0.any() and a1.any()
    assert got[0][1]["gldm_acc"].sum() + got[1][1]["gldm_acc"].sum() == int(mask.sum())
    assert 0 < got[0][1]["gldm_acc"].sum() < int(mask.sum())
