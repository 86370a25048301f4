"""Batch mode on N GPUs: cases (or voxel-map slabs) are independent, so they are sharded over the ranks of one
node with NO data-path collective -- the MI355X replacement for the reference's `multiprocessing.Pool` over cases
(radiomics/scripts/__init__.py:387-416).  One process per GPU (torch.distributed launch); results are small Python
objects and travel back to rank 0 with gather_object on the control plane only."""
from __future__ import annotations

import os
from typing import Callable, Sequence


def rank_world():
    """(rank, world) from torch.distributed when initialised, else from the launcher's environment, else (0, 1)"""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_indices(n_items: int, rank: int, world: int):
    """round-robin: case i goes to rank i % world (keeps long and short cases mixed on every GPU)"""
    return list(range(rank, n_items, world))


def split_slabs(n: int, parts: int, halo: int = 0):
    """[(lo, hi, lo_with_halo, hi_with_halo)] contiguous slabs of range(n) for voxel-map sharding: rank r computes
    centres lo..hi-1 and needs voxels lo_h..hi_h-1 (kernelRadius halo) of the replicated / host volume"""
    out = []
    base, extra = divmod(n, parts)
    lo = 0
    for r in range(parts):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi, max(0, lo - halo), min(n, hi + halo)))
        lo = hi
    return out


class _WorkerPool:
    """Persistent host threads for one GPU.  The native library keeps its context (workspaces, pinned buffers, lane
    streams, events) per host thread, so the threads must outlive a run_batch call: fresh threads per call would leave
    their contexts' device memory behind when they exit (hundreds of MB per 256^3 case) and would re-allocate every
    workspace inside the next call.  A thread that does end (shutdown_pools, interpreter exit) frees its workspace."""

    def __init__(self, threads: int, device):
        import queue
        import threading
        self.jobs = queue.Queue()
        self.device = device
        self.closed = False
        self.state = threading.Lock()          # guards `closed` against a run() racing a shutdown()
        self.threads = [threading.Thread(target=self._loop, daemon=True) for _ in range(threads)]
        for t in self.threads:
            t.start()

    def _loop(self):
        ctx = None
        try:
            import torch
        except ImportError:
            torch = None
        if torch is not None and self.device is not None:
            torch.cuda.set_device(self.device)      # torch's current device is per thread: stay on this rank's GPU
            ctx = torch.cuda.stream(torch.cuda.Stream())
            ctx.__enter__()
        try:
            while True:
                job = self.jobs.get()
                if job is None:
                    return
                job()
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
                try:                                  # this thread's native context dies with it: give its HBM back
                    from . import engine
                    engine.release_workspace()
                except Exception:
                    pass

    def run(self, indices, cases, worker):
        import threading
        out, errors = {}, []
        left = [len(indices)]
        lock, done = threading.Lock(), threading.Event()
        if not indices:
            return out

        def make(i):
            def job():
                try:
                    if not errors:
                        out[i] = worker(cases[i])
                except BaseException as e:          # surfaces in the caller's thread
                    errors.append(e)
                finally:
                    with lock:
                        left[0] -= 1
                        if left[0] == 0:
                            done.set()
            return job

        with self.state:
            # a pool that was retired by another host thread (a call with a different thread count on the same GPU) has no
            # workers left: jobs put on its queue would never run and done.wait() would block forever
            if self.closed:
                raise RuntimeError("worker pool was shut down (another call changed the thread count of this GPU's pool)")
            for i in indices:
                self.jobs.put(make(i))
        done.wait()
        if errors:
            raise errors[0]
        return out

    def run_many(self, indices, cases, many):
        """like run(), for a worker that overlaps consecutive cases: `many(iterable of cases)` is a generator of their results
        in order (RadiomicsFeatureExtractor.executeMany); every thread of the pool drives one such generator fed from a shared
        queue of the indices, so each thread has one case of look-ahead of its own"""
        import queue
        todo = queue.SimpleQueue()
        for i in indices:
            todo.put(i)
        out = {}

        def job(_):
            taken = []

            def feed():
                while True:
                    try:
                        i = todo.get_nowait()
                    except queue.Empty:
                        return
                    taken.append(i)
                    yield cases[i]
            for k, result in enumerate(many(feed())):
                out[taken[k]] = result

        n = len(self.threads)
        self.run(list(range(n)), [None] * n, job)
        return out

    def each(self, fn):
        """runs fn() once on EVERY thread of the pool (warm-up: each thread allocates its own native workspace)"""
        import threading
        barrier = threading.Barrier(len(self.threads))

        def job(_):
            barrier.wait()          # a thread that took one of these jobs cannot take a second one
            return fn()

        return self.run(list(range(len(self.threads))), [None] * len(self.threads), job)

    def shutdown(self):
        with self.state:
            if self.closed:
                return
            self.closed = True
            for _ in self.threads:           # (behind any jobs already queued: those still run)
                self.jobs.put(None)
        for t in self.threads:
            t.join()


_POOLS: dict = {}
_POOLS_LOCK = __import__("threading").Lock()


_ATEXIT: list = []


def shutdown_pools() -> None:
    """ends the worker threads of every pool (each frees the native workspace it held)"""
    with _POOLS_LOCK:
        pools = [_POOLS.pop(key) for key in list(_POOLS)]
    for p in pools:
        p.shutdown()


def _pool(threads: int):
    try:
        import torch
        dev = torch.cuda.current_device() if torch.cuda.is_available() else None
    except ImportError:
        dev = None
    key = (dev, max(1, threads))
    retired = []
    with _POOLS_LOCK:
        if key not in _POOLS:
            if not _ATEXIT:
                import atexit
                atexit.register(shutdown_pools)
                _ATEXIT.append(True)
            # one pool per GPU: a call with another thread count retires the old pool (its threads free the native workspaces
            # they held -- hundreds of MB per worker at 256^3) instead of keeping both alive; a host thread that still holds
            # the retired pool gets a RuntimeError from its next run(), not a hang
            retired = [_POOLS.pop(k) for k in [k for k in _POOLS if k[0] == dev]]
            _POOLS[key] = _WorkerPool(key[1], dev)
        pool = _POOLS[key]
    for old in retired:          # (outside the lock: joins the old threads, which finish the jobs they already hold)
        old.shutdown()
    return pool


def warm_threads(fn, threads: int):
    """fn() once on each of the `threads` worker threads run_batch(threads=) will use on this GPU"""
    return _pool(threads).each(fn)


def _run_threaded(indices, cases, worker, threads: int):
    """`threads` cases of this rank in flight on the one GPU: the library keeps its context (workspaces, streams, error
    state) per host thread and its C calls release the GIL, so one case's Python / launch overhead (a third of a 256^3
    case) runs under another case's kernels.  Every thread works on a HIP stream of its own; the threads persist
    between calls (_WorkerPool)."""
    for attempt in (0, 1):
        try:
            return _pool(threads).run(list(indices), cases, worker)
        except RuntimeError as e:      # (another host thread changed the pool's thread count between _pool() and run(): ADVICE r5)
            if attempt or "worker pool was shut down" not in str(e):
                raise


def run_batch(cases: Sequence, worker: Callable, gather: bool = True, threads: int = 1, many: Callable = None):
    """Applies `worker(case)` to this rank's shard.  With gather=True rank 0 returns the results of ALL cases in
    input order (other ranks return None); with gather=False every rank returns {case_index: result} of its own.
    threads > 1: that many cases of the shard at a time (host threads sharing the rank's GPU, see _run_threaded).
    `many(iterable of cases)` -> generator of their results in order, when given, replaces `worker`: a worker that overlaps
    consecutive cases on one thread (RadiomicsFeatureExtractor.executeMany: case i + 1's upload and filters under case i's
    last kernels), one such generator per host thread."""
    rank, world = rank_world()
    idx = shard_indices(len(cases), rank, world)
    if many is not None:
        mine = (_pool(threads).run_many(list(idx), cases, many) if threads > 1
                else dict(zip(idx, many(cases[i] for i in idx))))
    else:
        mine = _run_threaded(idx, cases, worker, threads) if threads > 1 else {i: worker(cases[i]) for i in idx}
    if not gather or world == 1:
        return [mine[i] for i in range(len(cases))] if (gather and world == 1) else mine
    import torch.distributed as dist
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(mine, bucket, dst=0)
    if rank != 0:
        return None
    merged = {}
    for part in bucket:
        merged.update(part)
    return [merged[i] for i in range(len(cases))]


def voxel_maps_sharded(feature_class, image, mask, features=None, **settings):
    """Voxel-based feature maps of one case on all ranks of the job: the centre-voxel list is cut into `world`
    contiguous slices (z-slabs in raster order), every rank evaluates its slice on its own GPU against the whole
    (replicated) volume, and rank 0 assembles the maps.  No data-path collective -- the per-rank results travel
    as Python objects on the control plane.  Returns {feature name: Image} on rank 0, None elsewhere.
    `feature_class` is one of the Radiomics* classes; `features` the names to enable (default: all)."""
    import numpy as np
    rank, world = rank_world()
    fc = feature_class(image, mask, voxelBased=True, voxelShard=(rank, world), **settings)
    for f in features or []:
        fc.enableFeatureByName(f)
    maps = fc.execute()
    coords = tuple(fc.labelledVoxelCoordinates)
    mine = {name: np.asarray(img.array)[coords] for name, img in maps.items()}
    if world == 1:
        return maps
    import torch.distributed as dist
    bucket = [None] * world if rank == 0 else None
    dist.gather_object((fc.labelledVoxelCoordinates, mine), bucket, dst=0)
    if rank != 0:
        return None
    out = {}
    for name, img in maps.items():
        full = np.array(img.array, copy=True)
        for c, vals in bucket:
            full[tuple(c)] = vals[name]
        out[name] = img.like(full)
    return out


# ---- one large segment over all ranks -------------------------------------------------------------------------
# The five matrices of ONE discretised volume, each split the way its arithmetic allows (third row):
#   GLCM, GLRLM   by ANGLE   the matrices are [.., .., Na] with independent angle columns; every rank sweeps its
#                            angles over the whole (replicated) volume -- runs never cross a rank boundary
#   GLDM, NGTDM   by z-SLAB  per-voxel histograms are additive: a rank accumulates the centre voxels of its planes
#                            (neighbours read from the planes around them) into integer accumulators
#   GLSZM         by LEVEL   a zone is a connected set of voxels of ONE level, so the zones of level g are those of
#                            the mask restricted to level g; rank r labels the levels g with g % world == r
# followed by ONE exchange step: all-reduce(sum) of the float64 count tensors (exact: integers and zeros) and of the
# int64 accumulators, and a gather of the compact zone tables.  The result is bit-identical to the single-GPU
# calls for every class.  The volume must be resident on every rank (replicate_volume broadcasts it over xGMI).

SEGMENT_CLASSES = ("glcm", "glrlm", "gldm", "ngtdm", "glszm")


class HipSegmentOps:
    """the rank-local pieces on this rank's GPU (pyradiomics_amd.engine; device tensors in, device tensors out)"""

    @staticmethod
    def pair_angles(shape, distances, force2D, force2Ddimension):
        from . import engine
        return engine.pair_angles(shape, distances, force2D, force2Ddimension)

    @staticmethod
    def neigh_angles(shape, distances, force2D, force2Ddimension):
        from . import engine
        return engine.neigh_angles(shape, distances, force2D, force2Ddimension)

    @staticmethod
    def pairs(image, mask, Ng, angles, force2D, force2Ddimension, fused_ok):
        """GLCM [Ng, Ng, na] of the given angles"""
        from . import engine
        if fused_ok:
            return engine.glcm_glrlm(image, mask, Ng, None, force2D, force2Ddimension, want_glrlm=False,
                                     angles=angles)[0]
        return engine.glcm(image, mask, Ng, (1,), force2D, force2Ddimension, angles=angles)[0]

    @staticmethod
    def runs(image, mask, Ng, Nr, angles, force2D, force2Ddimension):
        """GLRLM [Ng, Nr, na] of the given angles"""
        from . import engine
        return engine.glcm_glrlm(image, mask, Ng, Nr, force2D, force2Ddimension, want_glcm=False, angles=angles)[1]

    @staticmethod
    def pairs_runs(image, mask, Ng, Nr, angles, force2D, force2Ddimension):
        """both, one sweep per angle"""
        from . import engine
        g, r, _ = engine.glcm_glrlm(image, mask, Ng, Nr, force2D, force2Ddimension, angles=angles)
        return g, r

    @staticmethod
    def neigh_accumulate(family, image, mask, Ng, z_lo, z_hi, alpha, distances, force2D, force2Ddimension):
        from . import engine
        return engine.neigh_accumulate(family, image, mask, Ng, z_lo, z_hi, alpha, distances, force2D,
                                       force2Ddimension)

    @staticmethod
    def neigh_finalize(family, acc):
        from . import engine
        return engine.neigh_finalize(family, acc)

    @staticmethod
    def zones(image, mask, Ng, Ns, force2D, force2Ddimension):
        """compact GLSZM (P float64 numpy [Ng, k], sizes int numpy [k] ascending)"""
        from . import engine
        P, sizes = engine.glszm_compact(image, mask, Ng, Ns, force2D, force2Ddimension)
        return P.cpu().numpy(), sizes


def replicate_volume(image, mask, src: int = 0, device=None):
    """Broadcasts a discretised volume from rank `src` to every rank (one RCCL broadcast over xGMI): levels
    travel as ONE byte per voxel (0 = outside the mask) when they fit, int32 otherwise.  `image` / `mask` are
    tensors on `src` and ignored (may be None) elsewhere.  Returns (image int32, mask uint8) on this rank."""
    import torch
    import torch.distributed as dist
    rank, world = rank_world()
    if world == 1:
        return image.to(torch.int32), mask.to(torch.uint8)
    head = [None]
    if rank == src:
        device = image.device
        levels = torch.where(mask.bool(), image, torch.zeros_like(image))
        lo, hi = int(levels.min()), int(levels.max())
        narrow = lo >= 0 and hi <= 255
        # a voxel inside the mask with level 0 is an input error the matrix calls must still see (IndexError): it
        # cannot be told apart from "outside" in the packed form, so it is refused -- on EVERY rank, through the
        # header, before anybody waits in the data broadcast
        bad = bool(((levels == 0) & mask.bool()).any())
        head = [(tuple(image.shape), narrow, bad)]
    dist.broadcast_object_list(head, src=src)
    shape, narrow, bad = head[0]
    if bad:
        raise IndexError("level 0 under the mask")
    dtype = torch.uint8 if narrow else torch.int32
    if rank == src:
        packed = levels.to(dtype).contiguous()
    else:
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu"
        packed = torch.empty(shape, dtype=dtype, device=device)
    dist.broadcast(packed, src=src)
    return packed.to(torch.int32), (packed != 0).to(torch.uint8)


def segment_partials(image, mask, Ng, rank, world, classes=SEGMENT_CLASSES, Nr=None, alpha=0, distances=(1,),
                     force2D=False, force2Ddimension=0, ops=HipSegmentOps):
    """Rank `rank`'s additive share of the matrices of one segment (no communication).  Returns a dict:
    "glcm" [Ng,Ng,Na] / "glrlm" [Ng,Nr,Na] float64 with only this rank's angle columns non-zero, "gldm_acc" /
    "ngtdm_acc" int64 [Ng, Na+1] of this rank's planes, "glszm" (P numpy [Ng,k], sizes numpy [k]) of this rank's
    levels, plus the angle tables."""
    import numpy as np
    import torch
    shape = tuple(image.shape)
    dist1 = [int(d) for d in distances] == [1]
    out = {}
    if "glcm" in classes or "glrlm" in classes:
        run_angles = ops.pair_angles(shape, (1,), force2D, force2Ddimension)
        pair_angles = run_angles if dist1 else ops.pair_angles(shape, distances, force2D, force2Ddimension)
        if Nr is None:
            Nr = int(max(shape))
        mine_r = list(range(rank, len(run_angles), world))
        mine_p = list(range(rank, len(pair_angles), world))
        if "glcm" in classes:
            out["glcm"] = torch.zeros((Ng, Ng, len(pair_angles)), dtype=torch.float64, device=image.device)
            out["glcm_angles"] = pair_angles
        if "glrlm" in classes:
            out["glrlm"] = torch.zeros((Ng, Nr, len(run_angles)), dtype=torch.float64, device=image.device)
            out["glrlm_angles"] = run_angles
        if "glcm" in classes and "glrlm" in classes and dist1:
            if mine_r:
                g, r = ops.pairs_runs(image, mask, Ng, Nr, run_angles[mine_r], force2D, force2Ddimension)
                out["glcm"][:, :, mine_r] = g
                out["glrlm"][:, :, mine_r] = r
        else:
            if "glcm" in classes and mine_p:
                out["glcm"][:, :, mine_p] = ops.pairs(image, mask, Ng, pair_angles[mine_p], force2D,
                                                      force2Ddimension, dist1)
            if "glrlm" in classes and mine_r:
                out["glrlm"][:, :, mine_r] = ops.runs(image, mask, Ng, Nr, run_angles[mine_r], force2D,
                                                      force2Ddimension)
    if "gldm" in classes or "ngtdm" in classes:
        if image.dim() != 3:
            raise NotImplementedError("the z-slab split of GLDM / NGTDM needs a 3-D volume")
        lo, hi, _, _ = split_slabs(shape[0], world)[rank]
        out["neigh_angles"] = ops.neigh_angles(shape, distances, force2D, force2Ddimension)
        if "gldm" in classes:
            out["gldm_acc"] = ops.neigh_accumulate(0, image, mask, Ng, lo, hi, alpha, distances, force2D,
                                                   force2Ddimension)
        if "ngtdm" in classes:
            out["ngtdm_acc"] = ops.neigh_accumulate(1, image, mask, Ng, lo, hi, 0, distances, force2D,
                                                    force2Ddimension)
    if "glszm" in classes:
        own = mask.bool() & ((image % world) == rank)
        Ns = int(own.sum())
        if Ns:
            out["glszm"] = ops.zones(image, own.to(mask.dtype), Ng, Ns, force2D, force2Ddimension)
        else:
            out["glszm"] = (np.zeros((Ng, 0)), np.zeros(0, dtype=np.intc))
    return out


def merge_zone_tables(tables, Ng):
    """compact GLSZM tables of disjoint level sets -> one table over the union of their zone sizes"""
    import numpy as np
    sizes = np.unique(np.concatenate([np.asarray(s, dtype=np.int64) for _, s in tables])) if tables else np.zeros(0)
    P = np.zeros((Ng, len(sizes)), dtype=np.float64)
    for Pr, sr in tables:
        if len(sr):
            P[:, np.searchsorted(sizes, sr)] += Pr
    return P, sizes.astype(np.intc)


def segment_matrices_sharded(image, mask, Ng, classes=SEGMENT_CLASSES, Nr=None, alpha=0, distances=(1,),
                             force2D=False, force2Ddimension=0, ops=HipSegmentOps):
    """The matrices of ONE segment computed by all ranks of the job (see the table above): every rank passes the
    same discretised volume (int levels, mask; device tensors for the HIP ops -- use replicate_volume when only
    one rank holds it) and every rank returns the same dict: "glcm" [Ng,Ng,Na], "glrlm" [Ng,Nr,Na], "gldm"
    [Ng,2Na+1], "ngtdm" [Ng,3] (float64 tensors in the single-GPU layouts), "glszm" (P numpy [Ng,k], sizes [k]),
    "glcm_angles" / "glrlm_angles".  Bit-identical to the single-GPU calls."""
    rank, world = rank_world()
    part = segment_partials(image, mask, Ng, rank, world, classes, Nr, alpha, distances, force2D,
                            force2Ddimension, ops)
    tables = [part["glszm"]] if "glszm" in part else []
    if world > 1:
        import torch.distributed as dist
        for key in ("glcm", "glrlm", "gldm_acc", "ngtdm_acc"):     # the exchange step
            if key in part:
                dist.all_reduce(part[key], op=dist.ReduceOp.SUM)
        if "glszm" in part:
            tables = [None] * world
            dist.all_gather_object(tables, part["glszm"])
    out = {k: part[k] for k in ("glcm", "glrlm", "glcm_angles", "glrlm_angles") if k in part}
    if "gldm_acc" in part:
        out["gldm"] = ops.neigh_finalize(0, part["gldm_acc"])
    if "ngtdm_acc" in part:
        out["ngtdm"] = ops.neigh_finalize(1, part["ngtdm_acc"])
    if tables:
        P, sizes = merge_zone_tables(tables, Ng)
        if not len(sizes):
            raise IndexError("Calculation of GLSZM Failed.")      # empty mask, as the single-GPU call
        out["glszm"] = (P, sizes)
    return out
