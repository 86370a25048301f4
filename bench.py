#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: Mvoxels/s for the GLCM+GLRLM matrix build of a 512^3 volume at
32 grey levels (full mask, distance 1, 13 angles), 1..8 GPUs.

One "step" = one pass of the hot path over one volume that is already resident in HBM as the boundary dtypes
(int32 levels + uint8 mask, 5 B/voxel): pack -> 13 angle sweeps -> float64 GLCM [32,32,13] + GLRLM [32,512,13]
in HBM.  With N > 1 GPUs every rank builds the matrices of its own volume (batch mode shards cases, no collective:
SURVEY.md section 8e), so scaling is "weak" and value = N * voxels * steps / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 512] [--dist uniform|smooth]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (plus `roofline` and, at N=1, `cpu_baseline`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

# Fabric (HBM + Infinity Cache) bytes per engine call at the default workload, from the committed PMC profile
# profiles/r02b_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE calibrated on the
# kernels' own access widths: x2 for pack_rows' 16 B/lane loads, /0.524 for sweep_fw's 8 B/lane loads -- one angle
# alone reads the 134 MB level volume exactly once and reports 70.4 MB): pack_rows 0.80 GB + 0.152 GB written,
# sweep_fw 1.36 GB (12 angles, 10.2 volume reads).  rocprof cannot run inside bench.py; the figure is only
# reported when the workload matches the profiled one.
PROFILED_TRAFFIC = {"workload": (512, 32, "uniform"), "bytes": 2.31e9, "kernel_bytes": 1.36e9, "source": "profiles/r02b_pmc.md"}
# Secondary rooflines of the dominant kernel (it is not HBM-bound): wave-instructions per launch from the same PMC pass,
# ceilings from scripts/microbench.hip on this GPU (profiles/r02a_microbench.log): a conflict-free ds_add_u32 retires every
# 4.1 cycles per CU, an independent integer VALU instruction every 2.65 cycles per SIMD (256 CUs x 4 SIMDs, 2.4 GHz).
PROFILED_INSTS = {"workload": (512, 32, "uniform"), "lds": 2.802e7, "valu": 1.399e8, "source": "profiles/r02b_pmc.md"}
LDS_ATOMIC_PEAK = 256 * 2.4e9 / 4.1       # ds_add wave-instructions / s
VALU_PEAK = 1024 * 2.4e9 / 2.65           # VALU wave-instructions / s
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
ALG_BYTES_PER_VOXEL = 5.0       # int32 level + uint8 mask, read once


def make_volume(size: int, levels: int, dist: str, seed: int, device) -> tuple[torch.Tensor, torch.Tensor]:
    """Synthetic uint16-range volume discretised to `levels` grey levels (binWidth 25 on [0, 25*levels)),
    full mask.  `uniform`: iid levels (worst case for histogram locality); `smooth`: low-pass noise (diagonal-heavy
    GLCM, long runs) -- SURVEY.md section 8d, inputs C2(i)/(ii)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    shape = (size, size, size)
    if dist == "uniform":
        raw = torch.randint(0, 25 * levels, shape, generator=g, device=device, dtype=torch.int32)
    else:
        f = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        k = [w / 16.0 for w in (1, 4, 6, 4, 1)]
        for _ in range(3):                       # separable binomial blur (zero-padded), applied 3x (sigma ~ 1.7 voxels)
            for ax in range(3):
                out = f * k[2]
                for o in (1, 2):
                    out.narrow(ax, 0, size - o).add_(f.narrow(ax, o, size - o), alpha=k[2 + o])
                    out.narrow(ax, o, size - o).add_(f.narrow(ax, 0, size - o), alpha=k[2 - o])
                f = out
        lo, hi = f.min(), f.max()
        raw = ((f - lo) / (hi - lo) * (25 * levels - 1)).to(torch.int32)
    image = (raw // 25 + 1).to(torch.int32).contiguous()       # binImage with binWidth 25, min 0
    mask = torch.ones(shape, dtype=torch.uint8, device=device)
    return image, mask


def cpu_baseline(image: torch.Tensor, mask: torch.Tensor, levels: int, target_voxels: int, gpu_glcm=None):
    """Times the reference's own C (oracle/_ref, kind 'reference') or, if that prebuilt file is absent, our C
    restatement (kind 'port') on a z-slab of the same volume, single-threaded like the reference."""
    from oracle import binding
    if not os.path.exists(binding.PORT_SO):
        binding.build()
    kind = "reference" if binding.have_ref() else "port"
    cpu = binding.ref() if kind == "reference" else binding.port()
    nz = max(2, min(image.shape[0], target_voxels // (image.shape[1] * image.shape[2])))
    img = image[:nz].cpu().numpy()
    msk = mask[:nz].cpu().numpy().astype(bool)
    Nr = int(max(img.shape))
    t0 = time.perf_counter()
    g = cpu.calculate_glcm(img, msk, [1], levels, False, 0)[0]
    r = cpu.calculate_glrlm(img, msk, levels, Nr, False, 0)[0]
    dt = time.perf_counter() - t0
    out = {"value": round(img.size / dt / 1e6, 3), "unit": "Mvoxels/s", "cores": 1, "kind": kind,
           "sample": "z-slab %dx%dx%d of the bench volume, calculate_glcm + calculate_glrlm, %.1f s"
                     % (img.shape[0], img.shape[1], img.shape[2], dt)}
    return out, (img, msk, g[0], r[0], Nr)


def measured_copy_bandwidth(device) -> float:
    """device-to-device copy of 1 GiB (read + write = 2 GiB moved), GB/s: the bandwidth a pure streaming kernel
    achieves on this GPU, reported next to the 8 TB/s datasheet peak"""
    n = 1 << 28
    src = torch.empty(n, dtype=torch.int32, device=device).fill_(1)
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize()
    return 5 * 2 * n * 4 / (a.elapsed_time(b) * 1e-3) / 1e9


def usable_cores() -> int:
    """cores this process may really use: the affinity mask, capped by the cgroup CPU quota of the container"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota|max> <period>"
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:     # cgroup v1
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline_all_cores(levels: int, size: int, nz: int = 24):
    """The only way the reference uses more than one core: one process per case (scripts/__init__.py:387-416).
    nproc single-threaded workers, each on its own nz x size x size slab, started together; aggregate Mvoxels/s."""
    import subprocess
    import sys
    nproc = usable_cores()
    root = os.path.dirname(os.path.abspath(__file__))
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", str(nz), str(size), str(size), str(levels), str(i)],
                              cwd=root, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for i in range(nproc)]
    try:
        for p in procs:
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("cpu worker failed to start")
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        vox = 0
        for p in procs:
            v, _ = p.stdout.readline().split()
            vox += int(v)
        dt = time.perf_counter() - t0
    finally:
        for p in procs:
            p.kill() if p.poll() is None else None
            p.wait()
    return {"value": round(vox / dt / 1e6, 2), "unit": "Mvoxels/s", "nproc": nproc,
            "sample": "%d processes x %dx%dx%d slab each, %.1f s wall" % (nproc, nz, size, size, dt)}


def mode_batch(device, rank: int, cases: int, fence):
    """north_star 'batched mode' (BASELINE config 5 in miniature): whole cases -- a 256^3 volume, ball ROI, Original +
    8 wavelet sub-bands, all six feature classes -- through RadiomicsFeatureExtractor.execute, `cases` per rank,
    no collective; PRAD_BATCH_THREADS (default 3) cases in flight per GPU (batch.run_batch(..., threads=)).  Returns (cases,
    seconds, features per case)."""
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    from pyradiomics_amd.image import Image
    N = 256
    params = {"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}}
    zz, yy, xx = np.ogrid[:N, :N, :N]
    roi = np.zeros((N, N, N), dtype=np.int16)
    roi[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1
    from pyradiomics_amd import batch
    threads = int(os.environ.get("PRAD_BATCH_THREADS", "3"))
    ex = RadiomicsFeatureExtractor(params)
    vols = [(make_volume(N, 32, "smooth", 1000 * rank + c, device)[0] * 25).cpu().numpy().astype(np.int16)
            for c in range(cases + 1)]
    out = ex.execute(Image(vols[0]), Image(roi))          # warm-up: code objects, workspace

    def one(c):
        return ex.execute(Image(vols[c]), Image(roi))

    if threads > 1:                                       # (every thread warms its own workspace)
        batch._run_threaded(list(range(threads)), [0] * threads, one, threads)
    fence()
    t0 = time.perf_counter()
    res = batch._run_threaded(list(range(cases)), list(range(1, cases + 1)), one, threads)
    fence()
    return cases, time.perf_counter() - t0, len(res[0])


def mode_voxel(device, rank: int, world: int, size: int, fence):
    """north_star 'voxel-based mode' (BASELINE config 4): GLCM JointEntropy map of a size^3 volume with the
    exampleVoxel.yaml window (force2D, kernelRadius 2), every voxel a kernel centre; the centre list is cut into
    z-slabs, one per rank (batch.voxel_maps_sharded's split), the volume is resident on every rank, no collective.
    Returns (kernels of this rank, seconds)."""
    from pyradiomics_amd import engine
    img, msk = make_volume(size, 32, "smooth", 0, device)
    z0, z1 = (size * rank) // world, (size * (rank + 1)) // world
    zz, yy, xx = torch.meshgrid(torch.arange(z0, z1, device=device, dtype=torch.int32),
                                torch.arange(size, device=device, dtype=torch.int32),
                                torch.arange(size, device=device, dtype=torch.int32), indexing="ij")
    vox = torch.stack([zz.reshape(-1), yy.reshape(-1), xx.reshape(-1)])
    del zz, yy, xx
    kw = dict(kernelRadius=2, force2D=True, force2Ddimension=0)
    engine.voxel_glcm_features(img, msk, 32, vox[:, :4096].contiguous(), ["JointEntropy"], **kw)
    fence()
    t0 = time.perf_counter()
    res = engine.voxel_glcm_features(img, msk, 32, vox, ["JointEntropy"], **kw)
    fence()
    dt = time.perf_counter() - t0
    assert bool(torch.isfinite(res["JointEntropy"]).all())
    return int(vox.shape[1]), dt


def host_boundary(image, mask, Ng: int, Nr: int):
    """the drop-in call itself: pageable host numpy arrays in, float64 matrices out (PCIe inclusive; never `value`)"""
    from pyradiomics_amd import cmatrices
    img_h, msk_h = image.cpu().numpy(), mask.cpu().numpy()
    cmatrices.calculate_glcm_glrlm(img_h[:8], msk_h[:8], Ng, Nr, False, 0)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        cmatrices.calculate_glcm_glrlm(img_h, msk_h, Ng, Nr, False, 0)
        best = min(best, time.perf_counter() - t0)
    return {"ms_per_call": round(best * 1e3, 3), "Mvoxels_s": round(img_h.size / best / 1e6, 1),
            "input_GBps": round(5.0 * img_h.size / best / 1e9, 2),
            "note": "cmatrices.calculate_glcm_glrlm on pageable numpy int32 + uint8 arrays, best of 3; inputs go "
                    "through the pinned staging ring (4 host threads), PCIe Gen5 x16"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--levels", type=int, default=32)
    ap.add_argument("--dist", choices=["uniform", "smooth"], default="uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-modes", action="store_true", help="skip the batch / voxel-based side figures")
    ap.add_argument("--no-host-boundary", action="store_true", help="skip the host-pointer (drop-in) call timing")
    ap.add_argument("--cpu-voxels", type=int, default=320 * 512 * 512,
                    help="voxels in the CPU baseline sample (default: a 320-slice slab, ~10 s on one core)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist_on = world > 1 or "RANK" in os.environ      # any torch.distributed launch (also a 1-rank one) takes the N>1 path
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from pyradiomics_amd import engine

    image, mask = make_volume(args.size, args.levels, args.dist, seed=rank, device=device)
    Ng, Nr = args.levels, args.size
    nvox = image.numel()
    glcm = glrlm = None
    outs = [[None, None] for _ in range(4)]     # deferred steps alternate between the library's lanes: one output set each
    nstep = 0

    def step(deferred=False):
        nonlocal glcm, glrlm, nstep
        o = outs[nstep % len(outs)]
        nstep += 1
        glcm, glrlm, _ = engine.glcm_glrlm(image, mask, Ng, Nr, out_glcm=o[0], out_glrlm=o[1], deferred=deferred)
        o[0], o[1] = glcm, glrlm

    step()                                     # (synchronous: the dispatch verdict is read back)
    assert engine.last_path() == "sweep", "bench must run the sweep kernels, got %s" % engine.last_path()
    for _ in range(args.warmup):               # warm-up steps run like the timed ones (deferred: both lanes allocate
        step(deferred=True)                    # their workspaces here, not inside the timed region)
    engine.deferred_status()

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    # Timed region: K steps ENQUEUED back to back (deferred mode: no host synchronisation inside a step; the library deals
    # consecutive volumes onto its two lanes = internal streams, so the pack kernel of one volume shares the GPU with the
    # sweep kernel of the previous one), bracketed by barrier + synchronize.
    lanes = int(os.environ.get("PRAD_LANES", "2"))
    fence()
    engine.timing_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(deferred=True)
    fence()
    elapsed = time.perf_counter() - t0
    engine.deferred_status()          # raises if any step saw levels outside [1, Ng] (none can: synthetic levels)
    assert engine.timing_calls() == args.steps
    overlapped_ms = {fam: engine.timing_ms(fam) for fam in ("pack", "sweep", "finalize")}
    engine.timing_end()
    # Kernel durations for the roofline: the same K deferred steps on ONE lane (the caller's stream), where a kernel has
    # the GPU to itself -- with two lanes the launches of consecutive volumes overlap in time and a launch's duration
    # (reported as overlapped_kernel_ms) no longer says how fast the kernel is.  HIP events on the launch stream.
    engine.set_lanes(1)
    fence()
    engine.timing_begin()
    t0s = time.perf_counter()
    for _ in range(args.steps):
        step(deferred=True)
    fence()
    serial_elapsed = time.perf_counter() - t0s
    engine.deferred_status()
    kernel_ms = {fam: engine.timing_ms(fam) for fam in ("pack", "sweep", "finalize")}
    device_ms = engine.timing_ms(None)
    engine.timing_end()
    engine.set_lanes(0)
    # the synchronous drop-in call (host waits for every volume and reads the status back), informational
    sync_steps = max(3, min(10, args.steps))
    fence()
    t1 = time.perf_counter()
    for _ in range(sync_steps):
        step()
    sync_ms = (time.perf_counter() - t1) / sync_steps * 1e3
    def max_over_ranks(x: float) -> float:
        if not dist_on:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = max_over_ranks(elapsed)
    modes = None
    if not args.no_modes:
        # the two sharded modes north_star names, each rank on its own share, barrier + max-over-ranks like the headline
        torch.cuda.empty_cache()
        nc, dt_b, nfeat = mode_batch(device, rank, 12, fence)
        dt_b = max_over_ranks(dt_b)
        nk, dt_v = mode_voxel(device, rank, world, args.size, fence)
        dt_v = max_over_ranks(dt_v)
        nk_all = nk
        if dist_on:
            t = torch.tensor([nk], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            nk_all = int(t.item())
        modes = {"batch": {"value": round(world * nc / dt_b, 2), "unit": "cases/s", "cases_per_rank": nc,
                           "features_per_case": nfeat,
                           "case": "256^3 int16 volume from host memory, ball ROI (38 %% of the box), Original + 8 wavelet "
                                   "sub-bands, six feature classes; %s cases in flight per GPU (host threads, "
                                   "batch.run_batch(threads=))" % os.environ.get("PRAD_BATCH_THREADS", "3")},
                 "voxel": {"value": round(nk_all / dt_v / 1e6, 2), "unit": "Mkernels/s", "kernels": nk_all,
                           "case": "%d^3 volume, GLCM JointEntropy map, exampleVoxel.yaml window (force2D, kernelRadius 2), "
                                   "every voxel a centre, centres split into z-slabs over the ranks" % args.size}}
        torch.cuda.empty_cache()

    # size-independent property checks on the full-size result (every ordered neighbour pair / every voxel counted)
    gl = glcm.sum(dim=(0, 1)).cpu().numpy()
    n = args.size
    expect_pairs = {0: n * n * (n - 1), 1: n * (n - 1) * (n - 1), 2: (n - 1) ** 3}
    ang = engine._build_angles(np.array(image.shape, dtype=np.intc), None, False, -1)
    nocheck = bool(os.environ.get("PRAD_BENCH_NOCHECK"))   # ablation libraries (scripts/ablate.sh) are wrong by design
    for a in range(ang.shape[0]):
        if nocheck:
            break
        assert int(gl[a]) == expect_pairs[int(np.count_nonzero(ang[a])) - 1], "GLCM pair count of angle %d" % a
    rl_vox = (glrlm * torch.arange(1, Nr + 1, device=device, dtype=torch.float64).view(1, Nr, 1)).sum(dim=(0, 1))
    assert nocheck or torch.all(rl_vox == float(nvox)), "GLRLM runs do not tile the volume"

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * nvox * args.steps / elapsed / 1e6
        sweep_ms = kernel_ms["sweep"] / args.steps
        pipe_ms = device_ms / args.steps
        alg_bytes = ALG_BYTES_PER_VOXEL * nvox
        achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9
        copy_gbps = measured_copy_bandwidth(device)
        out = {
            "metric": "Mvoxels/s for GLCM+GLRLM build, %d^3 vol @%d bins" % (args.size, args.levels),
            "value": round(value, 1), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "sync_call_ms_per_step": round(sync_ms, 4),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 levels / u32 counts / f64 out", "data": "synthetic",
            "modes": modes,
            "config": {"workload": "GLCM+GLRLM matrix build, %d^3 int32+uint8 volume resident in HBM, %d grey levels, "
                                   "full mask, 13 angles, %s levels; one volume per GPU (batch sharding, no collective)"
                                   % (args.size, args.levels, args.dist),
                       "size": args.size, "levels": args.levels, "dist": args.dist},
            "roofline": {
                "bound": "hbm", "kernel": "sweep_fw_kernel (12 of the 13 angles; the x angle is walked inside "
                                          "pack_rows_fw_kernel, see pipeline_*)",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": PROFILED_TRAFFIC["bytes"] if (args.size, args.levels, args.dist) == PROFILED_TRAFFIC["workload"] else None,
                "traffic_source": PROFILED_TRAFFIC["source"] + " (fabric bytes of pack + sweeps per volume; traffic_kernel: the sweep kernel's share)",
                "traffic_kernel": PROFILED_TRAFFIC["kernel_bytes"] if (args.size, args.levels, args.dist) == PROFILED_TRAFFIC["workload"] else None,
                "algorithmic_bytes": alg_bytes, "kernel_ms": round(sweep_ms, 4),
                "pipeline_ms": round(pipe_ms, 4), "pack_ms": round(kernel_ms["pack"] / args.steps, 4),
                "finalize_ms": round(kernel_ms["finalize"] / args.steps, 4),
                "pipeline_achieved": round(alg_bytes / (pipe_ms * 1e-3) / 1e9, 2),
                "measured_copy_GBps": round(copy_gbps, 1), "frac_of_measured_copy": round(achieved / copy_gbps, 5),
                "lanes": lanes, "serial_ms_per_step": round(serial_elapsed / args.steps * 1e3, 4),
                "overlapped_kernel_ms": round(overlapped_ms["sweep"] / args.steps, 4),
                "overlapped_pack_ms": round(overlapped_ms["pack"] / args.steps, 4),
                "job_achieved": round(alg_bytes / (ms * 1e-3) / 1e9, 2), "job_frac": round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                "note": "achieved = 5 B/voxel x voxels / sweep kernel duration, HIP events on the launch stream over a second "
                        "K-step deferred loop on ONE lane (kernel_ms, pack_ms, finalize_ms, pipeline_*: launches do not "
                        "overlap there); the timed region that gives `value` runs the volumes on `lanes` lanes, where "
                        "launches of consecutive volumes share the GPU (overlapped_*: duration of a launch there); "
                        "job_* = 5 B/voxel over ms_per_step",
            },
        }
        if (args.size, args.levels, args.dist) == PROFILED_INSTS["workload"]:
            t = sweep_ms * 1e-3
            out["roofline"]["secondary"] = {
                "bound": "lds_atomic", "achieved": round(PROFILED_INSTS["lds"] / t / 1e9, 2), "peak": round(LDS_ATOMIC_PEAK / 1e9, 2),
                "unit": "G ds_add wave-instructions/s", "frac": round(PROFILED_INSTS["lds"] / t / LDS_ATOMIC_PEAK, 4),
                "note": "one LDS atomic per run end (1.11 per voxel-step on iid levels); on random bins the LDS array needs "
                        "6.5 cycles per instruction (70 % bank-conflict cycles), and each ds_add holds its SIMD ~8.8 cycles",
                "source": PROFILED_INSTS["source"]}
            out["roofline"]["secondary_valu"] = {
                "bound": "valu", "achieved": round(PROFILED_INSTS["valu"] / t / 1e9, 2), "peak": round(VALU_PEAK / 1e9, 2),
                "unit": "G VALU wave-instructions/s", "frac": round(PROFILED_INSTS["valu"] / t / VALU_PEAK, 4),
                "note": "5.5 VALU per voxel-step (4 in the exec-masked step, SDWA forms at ~4 cycles each)",
                "source": PROFILED_INSTS["source"]}
        if world == 1 and not args.no_host_boundary:
            out["host_boundary"] = host_boundary(image, mask, Ng, Nr)
        if world == 1 and not args.no_cpu_baseline:
            cb, (img, msk, g_cpu, r_cpu, Nr_s) = cpu_baseline(image, mask, Ng, args.cpu_voxels)
            # same slab through the GPU path: the CPU run doubles as a bit-exact parity check
            gg, rr, _ = engine.glcm_glrlm(image[:img.shape[0]].contiguous(), mask[:img.shape[0]].contiguous(), Ng, Nr_s)
            cb["parity"] = bool(np.array_equal(gg.cpu().numpy(), g_cpu) and np.array_equal(rr.cpu().numpy(), r_cpu))
            assert cb["parity"], "GPU matrices differ from the CPU baseline on the sample"
            try:                                   # informational: every host core busy, one process per volume
                cb["all_cores"] = cpu_baseline_all_cores(Ng, args.size)
            except Exception as e:                 # never let the side figure break the bench line
                cb["all_cores"] = {"error": str(e)[:200]}
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
