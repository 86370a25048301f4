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


def run_batch(cases: Sequence, worker: Callable, gather: bool = True):
    """Applies `worker(case)` to this rank's shard.  With gather=True rank 0 returns the results of ALL cases in
    input order (other ranks return None); with gather=False every rank returns {case_index: result} of its own."""
    rank, world = rank_world()
    mine = {i: worker(cases[i]) for i in shard_indices(len(cases), rank, world)}
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
