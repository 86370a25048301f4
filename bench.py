#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: Mvoxels/s for the GLCM+GLRLM matrix build of a 512^3 volume at
32 grey levels (full mask, distance 1, 13 angles), 1..8 GPUs.

One "step" = one pass of the hot path over one volume that is already resident in HBM as the boundary dtypes
(int32 levels + uint8 mask, 5 B/voxel): pack -> 13 angle walks -> float64 GLCM [32,32,13] + GLRLM [32,512,13]
in HBM.  With N > 1 GPUs every rank builds the matrices of its own volume (batch mode shards cases, no collective:
SURVEY.md section 8e), so scaling is "weak" and value = N * voxels * steps / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 512] [--dist uniform|smooth]

`--gpus N` with N > 1 re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1` (one rank per GPU over RCCL) unless it already runs under such a launch.

Rank 0 prints ONE JSON line: the headline (`value`, `roofline`), every other BASELINE config under `modes` (smooth
headline variant, config 2 five matrices 256^3, config 3 filter stack 256^3, config 4 voxel maps 2-D and 3-D windows,
config 5 batch of whole cases) and, at N = 1, `cpu_baseline` + `host_boundary`.
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

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
ALG_BYTES_PER_VOXEL = 5.0       # int32 level + uint8 mask, read once
# Secondary rooflines of the dominant kernel (it is not HBM-bound): ceilings from scripts/microbench.hip on this GPU
# (profiles/r02a_microbench.log): a conflict-free ds_add_u32 retires every 4.1 cycles per CU, an independent integer VALU
# instruction every 2.65 cycles per SIMD (256 CUs x 4 SIMDs, 2.4 GHz).
LDS_ATOMIC_PEAK = 256 * 2.4e9 / 4.1       # ds_add wave-instructions / s
VALU_PEAK = 1024 * 2.4e9 / 2.65           # VALU wave-instructions / s
# Counter figures that rocprofv3 collects (it cannot run inside bench.py): written by scripts/prof_r06.sh into this file
# from --pmc passes over THIS bench command with the committed build; used only when the workload matches.
PROFILED_FILE = os.path.join(ROOT, "profiles", "r06_counters.json")


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


def cpu_baseline_full(engine, image, mask, Ng, Nr, args, timed_outputs):
    """`cpu_baseline` of the JSON line.  The reference's own C (oracle/_ref; our restatement if that file is absent) on
    the WHOLE bench volume, its 13 angles dealt to the host's cores one core call per (matrix, angle) -- the core takes
    the angle table as an argument (cmatrices.c:4-92, :299-541) -- and compared bit for bit with (a) the matrices the
    timed loop left behind (deferred pipeline) and (b) a synchronous call on the same volume.  `value` is that run
    (cores = threads used); `one_core` is the single-threaded figure on a z-slab, as the reference runs a case;
    `all_cores` the reference's own way of using every core (one process per case)."""
    from oracle import binding
    if not os.path.exists(binding.PORT_SO):
        binding.build()
    kind = "reference" if binding.have_ref() else "port"
    cpu = binding.ref() if kind == "reference" else binding.port()
    img, msk = image.cpu().numpy(), mask.cpu().numpy().astype(bool)
    g_cpu, r_cpu, _ang, info = cpu.glcm_glrlm_angle_sharded(img, msk, Ng, Nr, usable_cores())
    g_def, r_def = (t.cpu().numpy() for t in timed_outputs)
    gs, rs, _ = engine.glcm_glrlm(image, mask, Ng, Nr)
    parity = {"deferred_pipeline": bool(np.array_equal(g_def, g_cpu) and np.array_equal(r_def, r_cpu)),
              "synchronous": bool(np.array_equal(gs.cpu().numpy(), g_cpu) and np.array_equal(rs.cpu().numpy(), r_cpu))}
    cb = {"value": round(img.size / info["wall_s"] / 1e6, 3), "unit": "Mvoxels/s", "cores": info["threads"], "kind": kind,
          "sample": "%dx%dx%d: the whole bench volume, calculate_glcm + calculate_glrlm as 26 (matrix, angle) core calls "
                    "over %d threads, %.1f s wall (%.1f s summed over the calls)"
                    % (img.shape + (info["threads"], info["wall_s"], info["cpu_s"])),
          "parity": all(parity.values()), "parity_detail": parity}
    assert cb["parity"], "GPU matrices differ from the reference C on the full bench volume: %s" % parity
    one, _ = cpu_baseline(image, mask, Ng, args.cpu_voxels)
    cb["one_core"] = one
    try:                                   # informational: every host core busy, one process per volume
        cb["all_cores"] = cpu_baseline_all_cores(Ng, args.size)
    except Exception as e:                 # never let the side figure break the bench line
        cb["all_cores"] = {"error": str(e)[:200]}
    return cb


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
    nproc = usable_cores()
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", str(nz), str(size), str(size), str(levels), str(i)],
                              cwd=ROOT, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for i in range(nproc)]
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


def frac_of_hbm(nbytes: float, ms: float) -> dict:
    """roofline entry of one stage: algorithmic bytes over its device time against the HBM peak"""
    gbps = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"ms": round(ms, 4), "algorithmic_bytes": int(nbytes), "achieved_GBps": round(gbps, 1),
            "frac": round(gbps / HBM_PEAK_GBPS, 5)}


def headline_loop(engine, image, mask, Ng, Nr, steps, warmup, fence, outs, families=True):
    """K deferred steps enqueued back to back (pipeline mode: the pack of volume N rides in the walk launch of volume
    N-1; deferred_join flushes the last volume INSIDE the timed region), bracketed by fence().  Inside the timed region only
    the launches of the dominant kernel are bracketed by HIP events (family "sweep", on the stream they are launched on):
    every event record costs the stream 3 - 6 us, and ten records per step -- what the library's full timing keeps --
    made a step 6 - 12 % slower than the product runs it (scripts/archive/r04_event_cost.py).  The other families (pack, x angle,
    finalize, whole call) come from a second, fully instrumented pass of the same loop AFTER the timed region.
    Returns (seconds, per-family device ms per step, last outputs)."""
    state = {"n": 0, "g": None, "r": None}

    def step(deferred=True):
        o = outs[state["n"] % len(outs)]
        state["n"] += 1
        g, r, _ = engine.glcm_glrlm(image, mask, Ng, Nr, out_glcm=o[0], out_glrlm=o[1], deferred=deferred)
        o[0], o[1] = g, r
        state["g"], state["r"] = g, r

    for _ in range(warmup):
        step()
    engine.deferred_status()
    fence()
    engine.timing_begin(only="sweep")
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    engine.deferred_join()            # the last volume's walks + finalize are enqueued here, before the closing fence
    fence()
    elapsed = time.perf_counter() - t0
    engine.deferred_status()          # raises if any step saw levels outside [1, Ng] (none can: synthetic levels)
    launches = engine.timing_count("sweep")
    assert launches >= steps, (launches, steps)          # one walk launch per volume (pipeline mode: exactly `steps`)
    fam = {"sweep": engine.timing_ms("sweep") / launches, "sweep_launches": launches}
    engine.timing_end()
    if families:                      # instrumented pass, outside the timed region
        k = max(3, min(steps, 10))
        engine.timing_begin()
        for _ in range(k):
            step()
        engine.deferred_join()
        fence()
        engine.deferred_status()
        for f in ("pack", "rows", "finalize", "sweep"):
            fam[f + "_instrumented" if f == "sweep" else f] = engine.timing_ms(f) / k
        fam["device"] = engine.timing_ms(None) / k
        engine.timing_end()
    else:
        fam.update({"pack": 0.0, "rows": 0.0, "finalize": 0.0, "device": 0.0, "sweep_instrumented": 0.0})
    return elapsed, fam, (state["g"], state["r"])


def mode_config2(device, engine, size=256, levels=32):
    """BASELINE config 2: all five texture matrices of one size^3 volume (uniform and smooth levels), device-resident,
    synchronous calls; device ms per matrix from the library's HIP events, 5 B/voxel algorithmic per matrix call."""
    out = {"case": "%d^3 int32+uint8 volume, %d grey levels, full mask; per matrix: device ms of the synchronous call "
                   "(HIP events), 5 B/voxel algorithmic" % (size, levels)}
    for dist in ("uniform", "smooth"):
        img, msk = make_volume(size, levels, dist, 0, device)
        n = img.numel()
        jobs = {"glcm_glrlm": lambda: engine.glcm_glrlm(img, msk, levels, size),
                "gldm": lambda: engine.gldm(img, msk, levels),
                "ngtdm": lambda: engine.ngtdm(img, msk, levels),
                "glszm": lambda: engine.glszm_compact(img, msk, levels, n)}
        res, total = {}, 0.0
        for name, fn in jobs.items():
            fn()
            best_dev, best_wall = 1e9, 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                best_wall = min(best_wall, (time.perf_counter() - t0) * 1e3)
                best_dev = min(best_dev, engine.last_device_ms())
            res[name] = dict(frac_of_hbm(ALG_BYTES_PER_VOXEL * n, best_dev), wall_ms=round(best_wall, 4), path=engine.last_path())
            total += best_wall
        res["total_wall_ms"] = round(total, 4)
        res["Mvoxels_s_all_five"] = round(n / (total * 1e-3) / 1e6, 1)
        # GLDM and NGTDM from ONE pass over the neighbourhoods (prad_calculate_gldm_ngtdm_dev): what the two feature classes
        # of a derived image share in the product route
        fn = lambda: engine.gldm_ngtdm(img, msk, levels, 0)
        fn()
        best_dev, best_wall = 1e9, 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best_wall = min(best_wall, (time.perf_counter() - t0) * 1e3)
            best_dev = min(best_dev, engine.last_device_ms())
        res["gldm_ngtdm_one_pass"] = dict(frac_of_hbm(ALG_BYTES_PER_VOXEL * n, best_dev), wall_ms=round(best_wall, 4))
        # THE PRODUCT ROUTE (featureextractor._startFeatures): every class of the image -- the five matrices AND their feature
        # formulas -- queued by one prad_image_enqueue_dev call on the library's side streams, one wait per image, the next
        # image queued before the previous one is collected.  Wall ms per image over a run of 8 images.
        classes = engine.IMG_GLCM | engine.IMG_GLRLM | engine.IMG_GLDM | engine.IMG_NGTDM | engine.IMG_GLSZM

        def enqueued_run(k):
            pending = None
            for _ in range(k):
                tok = engine.image_enqueue(img, msk, None, levels, n, classes)
                if pending is not None:
                    assert engine.image_wait(pending)
                pending = tok
            assert engine.image_wait(pending)
        enqueued_run(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enqueued_run(8)
        torch.cuda.synchronize()
        enq = (time.perf_counter() - t0) * 1e3 / 8
        res["enqueued_ms"] = round(enq, 4)
        res["enqueued"] = dict(frac_of_hbm(ALG_BYTES_PER_VOXEL * n, enq),
                               what="all five matrices + their feature formulas of one image, one library call, wall per image")
        out[dist] = res
    return out


def mode_fallback(device, engine, size=256):
    """what the sweeps do not take: distances [1, 2], 161+ grey levels.  Until round 4 these calls ran on the exact generic
    kernels (fp64 L2 atomics: 43 - 85 ms); round 5 put the pairs tier (kernels_pairs.h: LDS pair tables per angle group, run
    starts walked, 16-bit levels) in between -- `path` says which one a call took.  Parity: tests/test_gpu_pairs.py,
    test_gpu_parity.py, test_gpu_fuzz.py."""
    g = torch.Generator(device=device)
    g.manual_seed(11)
    shape = (size, size, size)
    img = torch.randint(1, 301, shape, generator=g, device=device, dtype=torch.int32)
    msk = torch.ones(shape, dtype=torch.uint8, device=device)
    n = img.numel()
    img32 = (img - 1) % 32 + 1
    img255 = (img - 1) % 255 + 1
    jobs = {"glcm_d12_Ng32": lambda: engine.glcm(img32, msk, 32, (1, 2)),
            "glcm_glrlm_Ng255": lambda: engine.glcm_glrlm(img255, msk, 255, size),
            "glcm_d12_Ng300": lambda: engine.glcm(img, msk, 300, (1, 2)),
            "glcm_glrlm_Ng300": lambda: engine.glcm_glrlm(img, msk, 300, size),
            "gldm_d12_Ng300": lambda: engine.gldm(img, msk, 300, 0, (1, 2)),
            "ngtdm_d12_Ng300": lambda: engine.ngtdm(img, msk, 300, (1, 2)),
            "glszm_Ng300": lambda: engine.glszm_compact(img, msk, 300, n)}
    res = {"case": "%d^3 int32 volume, 300 iid grey levels, full mask, distances [1, 2] where the class takes distances; "
                   "device ms of the synchronous call" % size}
    for name, fn in jobs.items():
        try:
            fn()
            best_dev, best_wall = 1e9, 1e9
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                best_wall = min(best_wall, (time.perf_counter() - t0) * 1e3)
                best_dev = min(best_dev, engine.last_device_ms())
            res[name] = dict(frac_of_hbm(ALG_BYTES_PER_VOXEL * n, best_dev), wall_ms=round(best_wall, 4), path=engine.last_path())
        except Exception as e:                  # noqa: BLE001
            res[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    return res


def mode_config3(device, engine, size=256):
    """BASELINE config 3: wavelet (coif1, 8 sub-bands) + LoG (sigma 1..5 mm) of a size^3 int16 volume, every derived
    image re-discretised to 32 levels (binCount) and pushed through GLCM+GLRLM; nothing leaves HBM.  Per-stage wall ms
    (synchronised per stage) with the stage's algorithmic bytes.  Filter parity: tests/test_notebook_pin.py."""
    lv, msk = make_volume(size, 32, "smooth", 0, device)
    img = (lv.to(torch.float32) * 25.0 + 3.0).to(torch.int16)
    n = img.numel()

    def run():
        t = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        derived = engine.wavelet_images(img)
        torch.cuda.synchronize()
        t["wavelet_x8"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        sig = (1.0, 2.0, 3.0, 4.0, 5.0)      # (the five sigmas in the same launches, as filters.getLoGImage runs them)
        for s, d in zip(sig, engine.log_images(img, (1.0, 1.0, 1.0), sig)):
            derived["log-sigma-%g" % s] = d
        torch.cuda.synchronize()
        t["log_x5"] = (time.perf_counter() - t0) * 1e3
        tb = tm = 0.0
        for name, d in derived.items():
            t0 = time.perf_counter()
            levels, Ng, _, _ = engine.bin_image(d, msk, with_counts=True, binCount=32)
            torch.cuda.synchronize()
            tb += (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            engine.glcm_glrlm(levels, msk, Ng, size)
            torch.cuda.synchronize()
            tm += (time.perf_counter() - t0) * 1e3
            assert engine.last_path() == "sweep"
        t["binning_x13"] = tb
        t["glcm_glrlm_x13"] = tm
        return t

    def run_product():
        """the route featureextractor.execute takes: filters, then per derived image binCount + ONE enqueue call for GLCM and
        GLRLM (matrices and feature formulas), the next image queued before the previous one is collected; one clock around
        everything"""
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        derived = engine.wavelet_images(img)
        sig = (1.0, 2.0, 3.0, 4.0, 5.0)
        for s, d in zip(sig, engine.log_images(img, (1.0, 1.0, 1.0), sig)):
            derived["log-sigma-%g" % s] = d
        pending = None
        for name, d in derived.items():
            levels, Ng, _, _ = engine.bin_image(d, msk, with_counts=True, binCount=32)
            tok = engine.image_enqueue(levels, msk, None, Ng, n, engine.IMG_GLCM | engine.IMG_GLRLM)
            if pending is not None:
                assert engine.image_wait(pending)
            pending = tok
        assert engine.image_wait(pending)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    run()
    t = run()
    run_product()
    product_ms = min(run_product() for _ in range(2))
    # algorithmic bytes: wavelet 8 B in + 8 x 8 B out per voxel (float64); LoG 4 B in + 4 B out per sigma (float32); binning
    # per image: min/max pass (dtype + 1) + digitize pass (dtype + 1 in, 4 out); matrices 5 B per voxel and image
    bin_bytes = 8 * (2 * 9 + 4) + 5 * (2 * 5 + 4)
    stages = {"wavelet_x8": frac_of_hbm(72.0 * n, t["wavelet_x8"]), "log_x5": frac_of_hbm(5 * 8.0 * n, t["log_x5"]),
              "binning_x13": frac_of_hbm(float(bin_bytes) * n, t["binning_x13"]),
              "glcm_glrlm_x13": frac_of_hbm(13 * 5.0 * n, t["glcm_glrlm_x13"])}
    total = sum(t.values())
    return {"case": "%d^3 int16 volume -> 8 wavelet sub-bands + 5 LoG images -> binCount 32 -> GLCM+GLRLM, wall ms per "
                    "stage (one synchronisation per stage / image)" % size,
            "stages": stages, "staged_total_ms": round(total, 3),
            "total_ms": round(product_ms, 3),
            "total_is": "the product route: filters, binCount, one enqueue call per derived image (GLCM + GLRLM matrices and "
                        "formulas), look-ahead of one image, one clock around the whole case; staged_total_ms = the sum of "
                        "the per-stage figures above (one synchronisation per stage / image)",
            "Mvoxels_s_derived": round(13 * n / (product_ms * 1e-3) / 1e6, 1)}


def _batch_case_setup(device, seed_base: int, ncases: int):
    """the extractor, ROI and synthetic volumes of batch mode (shared by the parent and its worker processes)"""
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    from pyradiomics_amd.image import Image
    N = 256
    params = {"setting": {"binCount": 32, "additionalInfo": False}, "imageType": {"Original": {}, "Wavelet": {}}}
    zz, yy, xx = np.ogrid[:N, :N, :N]
    roi = np.zeros((N, N, N), dtype=np.int16)
    roi[((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2] = 1
    ex = RadiomicsFeatureExtractor(params)
    vols = [(make_volume(N, 32, "smooth", seed_base + c, device)[0] * 25).cpu().numpy().astype(np.int16) for c in range(ncases)]
    one = lambda c: ex.execute(Image(vols[c]), Image(roi))                                   # noqa: E731
    # consecutive cases of one thread with one case of overlap (executeMany); PRAD_BATCH_PIPELINE=0: case after case
    one.many = (lambda cs: ex.executeMany((Image(vols[c]), Image(roi)) for c in cs)) if os.environ.get("PRAD_BATCH_PIPELINE", "1") != "0" else None
    return ex, vols, one


def batch_child(argv) -> None:
    """one worker PROCESS of batch mode: `bench.py --batch-child <device index> <cases> <threads> <seed base>`.  Builds its
    cases, warms up, prints `ready`, waits for a line on stdin (all workers start together), runs its cases through
    RadiomicsFeatureExtractor.execute and prints `<cases> <seconds> <features per case>`"""
    from pyradiomics_amd import batch
    dev_index, ncases, threads, seed = (int(a) for a in argv[:4])
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    ex, vols, one = _batch_case_setup(device, seed, ncases + 1)
    one(0)
    one(0)
    if threads > 1:
        batch.warm_threads(lambda: one(0), threads)
    torch.cuda.synchronize()
    print("ready", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    res = _batch_run(batch, one, ncases, threads)
    torch.cuda.synchronize()
    print("%d %.6f %d" % (ncases, time.perf_counter() - t0, len(res[0])), flush=True)


def _batch_run(batch, one, ncases: int, threads: int):
    """cases 1 .. ncases of a worker (process or thread pool): {index: features}"""
    ids = list(range(1, ncases + 1))
    if one.many is not None:
        if threads > 1:
            return batch._pool(threads).run_many(list(range(ncases)), ids, one.many)
        return dict(enumerate(one.many(ids)))
    if threads > 1:
        return batch._run_threaded(list(range(ncases)), ids, one, threads)
    return {i: one(i + 1) for i in range(ncases)}


def mode_batch(device, rank: int, cases: int, fence, world: int = 1):
    """north_star 'batched mode' (BASELINE config 5 in miniature): whole cases -- a 256^3 volume, ball ROI, Original +
    8 wavelet sub-bands, all six feature classes -- through RadiomicsFeatureExtractor, `cases` per rank, no collective.
    Layout (round 6): with at least five host cores per rank, PRAD_BATCH_PROCS = 4 worker processes on the rank's GPU (what rounds
    3 - 5 reported, and what the reference does with multiprocessing.Pool over cores, scripts/__init__.py:387-416), each with one
    case of overlap (executeMany); with fewer, ONE process: PRAD_BATCH_THREADS = 3 host threads
    (batch.run_batch(threads=3, many=executeMany)), each with a launcher thread of its own for the ~65 launches of a derived image.
    PRAD_BATCH_PROCS = N forces N worker processes (0: threads of this process), capped at usable_cores() // world.
    Returns (cases, seconds, features per case)."""
    import subprocess
    from pyradiomics_amd import batch
    # round 6: a rank whose share of the host is fewer than five cores runs ONE process -- three host threads, each with a launcher
    # thread of its own (prad_image_submit) and one case of overlap (executeMany): 90 - 100 % of what four worker processes reach
    # (profiles/r06_probes.md section 6) on ~2 cores, so eight ranks fit the 16 cores of the bench host; with the cores to spare
    # (one rank on that host) four worker processes stay the default: no GIL, no shared runtime locks, the GPU's own rate
    auto_procs = "4" if usable_cores() // max(1, world) >= 5 else "0"
    procs = int(os.environ.get("PRAD_BATCH_PROCS", auto_procs))
    if procs > 0:
        # every worker process is a host-bound Python thread: more workers than cores only take turns.  The ranks of a node
        # share the host (N ranks x 4 workers + N parents), so a rank gets usable_cores() // world of them (VERDICT r4 missing #5)
        procs = max(1, min(procs, usable_cores() // max(1, world)))
    threads = int(os.environ.get("PRAD_BATCH_THREADS", "1" if procs > 0 else "3"))
    ex, vols, one = _batch_case_setup(device, 1000 * rank, min(cases, 5) + 1 if procs > 0 else cases + 1)
    one(0)                                                # warm-up: code objects, workspace
    if procs > 0:
        per = [cases // procs + (1 if i < cases % procs else 0) for i in range(procs)]
        per = [p for p in per if p > 0]
        kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--batch-child", str(device.index or 0), str(p),
                                  str(threads), str(1000 * rank + 100 * (i + 1))], cwd=ROOT, stdin=subprocess.PIPE,
                                 stdout=subprocess.PIPE, text=True) for i, p in enumerate(per)]
        try:
            for k in kids:
                line = k.stdout.readline().strip()
                if line != "ready":
                    raise RuntimeError("batch worker failed to start: %r" % line)
            fence()
            t0 = time.perf_counter()
            for k in kids:
                k.stdin.write("go\n")
                k.stdin.flush()
            done, nfeat = 0, 0
            for k in kids:
                n, _, nf = k.stdout.readline().split()
                done += int(n)
                nfeat = int(nf)
            fence()
            dt = time.perf_counter() - t0
        finally:
            for k in kids:
                if k.poll() is None:
                    try:
                        k.wait(timeout=30)
                    except subprocess.TimeoutExpired:
                        k.kill()
        assert done == cases
        mode_batch.how = "%d worker processes x %d thread(s) on the GPU%s" % (len(per), threads, ", one case of overlap per thread (executeMany)" if one.many else "")
    else:
        if threads > 1:                                   # every worker thread warms its own workspace
            batch.warm_threads(lambda: one(0), threads)
        fence()
        t0 = time.perf_counter()
        res = _batch_run(batch, one, cases, threads)
        fence()
        dt = time.perf_counter() - t0
        nfeat = len(res[0])
        mode_batch.how = "%d host thread(s) of one process (batch.run_batch(threads=%s))" % (threads, ", many=executeMany" if one.many else "")
    lat = []                                              # one case at a time on this thread (the case pipeline's latency)
    for c in range(1, min(cases, 5) + 1):
        fence()
        t1 = time.perf_counter()
        one(c)
        fence()
        lat.append((time.perf_counter() - t1) * 1e3)
    mode_batch.one_thread_ms = sorted(lat)[len(lat) // 2]
    # ... and the same cases back to back on this one thread (no fence between them, one case of overlap when executeMany is on):
    # what a single-threaded caller that loops over its cases gets per case
    loop_cases = list(range(1, min(cases, 5) + 1)) * 2
    fence()
    t1 = time.perf_counter()
    if one.many is not None:
        for _ in one.many(loop_cases):
            pass
    else:
        for c in loop_cases:
            one(c)
    fence()
    mode_batch.one_thread_loop_ms = (time.perf_counter() - t1) / len(loop_cases) * 1e3
    return cases, dt, nfeat


def mode_voxel(device, rank: int, world: int, size: int, fence, three_d: bool = False):
    """north_star 'voxel-based mode' (BASELINE config 4): GLCM JointEntropy map of a size^3 volume, every voxel a kernel
    centre -- the exampleVoxel.yaml window (force2D, kernelRadius 2: 5 x 5) or, with three_d, the 3-D 5^3 window; the
    centre list is cut into z-slabs, one per rank (batch.voxel_maps_sharded's split), the volume is resident on every
    rank, no collective.  Returns (kernels of this rank, seconds, kernel device ms)."""
    from pyradiomics_amd import engine
    img, msk = make_volume(size, 32, "smooth", 0, device)
    z0, z1 = (size * rank) // world, (size * (rank + 1)) // world
    zz, yy, xx = torch.meshgrid(torch.arange(z0, z1, device=device, dtype=torch.int32),
                                torch.arange(size, device=device, dtype=torch.int32),
                                torch.arange(size, device=device, dtype=torch.int32), indexing="ij")
    vox = torch.stack([zz.reshape(-1), yy.reshape(-1), xx.reshape(-1)])
    del zz, yy, xx
    kw = dict(kernelRadius=2, force2D=not three_d, force2Ddimension=0)
    # warm-up with the DENSE request itself: a sparse one takes the window kernel and leaves the sliding-window route's
    # workspace (the whole-volume maps, ~1.6 GB at 512^3) to be allocated inside the timed call -- VERDICT r4 weak #2: the
    # figure swung 7x between runs with the same kernel time
    res = engine.voxel_glcm_features(img, msk, 32, vox, ["JointEntropy"], **kw)
    assert engine.last_variant() == "slide", engine.last_variant()
    del res
    fence()
    dts = []
    for _ in range(2):               # two consecutive timed calls; the slower one is reported, both are kept
        t0 = time.perf_counter()
        res = engine.voxel_glcm_features(img, msk, 32, vox, ["JointEntropy"], **kw)
        fence()
        dts.append(time.perf_counter() - t0)
    kms = engine.last_kernel_ms("voxel")
    assert bool(torch.isfinite(res["JointEntropy"]).all())
    mode_voxel.runs_ms = [round(d * 1e3, 3) for d in dts]
    return int(vox.shape[1]), max(dts), kms


def mode_voxel_wide(device, engine, size=256):
    """the sliding-window kernel beyond JointEntropy (round 5): all seventeen features it carries as maps of a size^3 volume,
    exampleVoxel.yaml window (force2D, kernelRadius 2), every voxel a centre; against the three-feature instantiation"""
    from pyradiomics_amd.cmatrices import VOXEL_GLCM_FEATURES
    img, msk = make_volume(size, 32, "smooth", 0, device)
    idx = torch.arange(size, device=device, dtype=torch.int32)
    zz, yy, xx = torch.meshgrid(idx, idx, idx, indexing="ij")
    vox = torch.stack([zz.reshape(-1), yy.reshape(-1), xx.reshape(-1)])
    del zz, yy, xx
    kw = dict(kernelRadius=2, force2D=True, force2Ddimension=0)
    not_carried = {"Correlation", "DifferenceEntropy", "SumEntropy", "Imc1", "Imc2", "MaximumProbability"}
    wide = [f for f in VOXEL_GLCM_FEATURES if f not in not_carried]
    out = {"case": "%d^3 volume, 5x5 window, every voxel a centre" % size}
    for name, feats in (("three_features", ["JointEntropy", "JointEnergy", "JointAverage"]), ("seventeen_features", wide)):
        engine.voxel_glcm_features(img, msk, 32, vox, feats, **kw)
        assert engine.last_variant() == "slide", (name, engine.last_variant())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engine.voxel_glcm_features(img, msk, 32, vox, feats, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {"features": len(feats), "call_ms": round(dt * 1e3, 3), "kernel_ms": round(engine.last_kernel_ms("voxel"), 3),
                     "Mkernels_s": round(vox.shape[1] / dt / 1e6, 1)}
    return out


def mode_voxel_brain1(device, engine):
    """the reference's own voxel example on the reference's own data: examples/exampleSettings/exampleVoxel.yaml (binWidth 25,
    force2D, kernelRadius 2, GLCM JointEntropy) on data/brain1 (tests/golden/data: 256 x 256 x 25 int16, 4137 ROI voxels, 33 grey
    levels after binning) -- the crop padded by kernelRadius, every ROI voxel a centre.  Wall ms of the map call (levels
    resident) and which kernel took it; parity of exactly this request: tests/test_gpu_configs.py."""
    from pyradiomics_amd import imageoperations
    from pyradiomics_amd.image import read_nrrd
    gd = os.path.join(ROOT, "tests", "golden", "data")
    image, mask = read_nrrd(os.path.join(gd, "brain1_image.nrrd")), read_nrrd(os.path.join(gd, "brain1_label.nrrd"))
    ci, cmk = imageoperations.cropToTumorMask(image, mask, 1, padDistance=2)
    roi = cmk.array == 1
    levels, _ = imageoperations.binImage(ci.array, roi, binWidth=25)
    levels = np.where(roi, levels, 0).astype(np.int32)
    Ng = int(levels.max())
    vox = torch.from_numpy(np.array(np.nonzero(roi)).astype(np.int32)).to(device)
    lv, mk = torch.from_numpy(levels).to(device), torch.from_numpy(roi.astype(np.uint8)).to(device)
    kw = dict(kernelRadius=2, force2D=True, force2Ddimension=0)
    engine.voxel_glcm_features(lv, mk, Ng, vox, ["JointEntropy"], **kw)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = engine.voxel_glcm_features(lv, mk, Ng, vox, ["JointEntropy"], **kw)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    assert bool(torch.isfinite(res["JointEntropy"]).all())
    kms, variant = engine.last_kernel_ms("voxel"), engine.last_variant()
    # cpu_baseline leg of this mode: the reference's route on one host core -- per-kernel matrices from the reference C
    # (_cmatrices.c:203-222), then glcm.py:149-205 / :560-576 in numpy -- on the same 4137 kernels
    from oracle import binding
    cpu = binding.ref() if binding.have_ref() else binding.port()
    vh = vox.cpu().numpy()
    t0 = time.perf_counter()
    ent = np.empty(vh.shape[1])
    for lo in range(0, vh.shape[1], 1000):
        P, _ = cpu.calculate_glcm(levels, roi, [1], Ng, True, 0, kernelRadius=2, voxels=np.ascontiguousarray(vh[:, lo:lo + 1000]))
        P = P + P.transpose(0, 2, 1, 3)
        tot = P.sum((1, 2))
        tot[tot == 0] = np.nan
        with np.errstate(invalid="ignore", divide="ignore"):
            p = P / tot[:, None, None, :]
            ent[lo:lo + 1000] = np.nanmean(-(p * np.log2(p + np.spacing(1))).sum((1, 2)), 1)
    cpu_s = time.perf_counter() - t0
    agree = bool(np.allclose(res["JointEntropy"].cpu().numpy(), ent, rtol=1e-9, atol=1e-12))
    return {"case": "exampleVoxel.yaml on brain1: crop %s, %d centres, %d grey levels" % (tuple(levels.shape), int(vox.shape[1]), Ng),
            "call_ms": round(best * 1e3, 4), "kernel_ms": round(kms, 4), "variant": variant,
            "Mkernels_s": round(int(vox.shape[1]) / best / 1e6, 3),
            "cpu_reference_route": {"kernels_s": round(vh.shape[1] / cpu_s, 1), "cores": 1, "seconds": round(cpu_s, 3),
                                    "kind": "reference" if binding.have_ref() else "port", "agrees_1e-9": agree}}


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


def respawn_under_torchrun(n: int) -> None:
    """`python bench.py --gpus N` (N > 1) outside a torch.distributed launch: become that launch"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main() -> None:
    if len(sys.argv) > 1 and sys.argv[1] == "--batch-child":
        batch_child(sys.argv[2:])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--levels", type=int, default=32)
    ap.add_argument("--dist", choices=["uniform", "smooth"], default="uniform")
    ap.add_argument("--device-warmup-ms", type=float, default=40.0,
                    help="untimed passes of the headline loop before the W warm-up steps (a process's first ~20 ms of GPU work run "
                         "6 - 7 %% slower than the steady state; 0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-modes", action="store_true", help="skip the side figures (smooth variant, configs 2-5)")
    ap.add_argument("--no-host-boundary", action="store_true", help="skip the host-pointer (drop-in) call timing")
    ap.add_argument("--deferred-mode", choices=["pipeline", "lanes"], default=None,
                    help="how deferred calls overlap (default: the library's, i.e. pipeline unless PRAD_DEFERRED_MODE=lanes)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo lets several ranks share "
                                                      "one GPU on a test box)")
    ap.add_argument("--batch-cases", type=int, default=36, help="cases per rank in modes.batch")
    ap.add_argument("--cpu-voxels", type=int, default=160 * 512 * 512,
                    help="voxels in the single-threaded CPU sample (default: a 160-slice slab, ~5 s on one core)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        respawn_under_torchrun(args.gpus)                 # does not return
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback")
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % ndev)              # (several ranks may share a GPU when a box has fewer: test setups)
    device = torch.device("cuda", local_rank % ndev)
    dist_on = world > 1 or "RANK" in os.environ      # any torch.distributed launch (also a 1-rank one) takes the N>1 path
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    from pyradiomics_amd import engine
    if args.deferred_mode:
        engine.set_deferred_mode(1 if args.deferred_mode == "pipeline" else 0)
    mode_name = args.deferred_mode or ("lanes" if os.environ.get("PRAD_DEFERRED_MODE") == "lanes" else "pipeline")

    image, mask = make_volume(args.size, args.levels, args.dist, seed=rank, device=device)
    Ng, Nr = args.levels, args.size
    nvox = image.numel()
    outs = [[None, None] for _ in range(4)]     # consecutive deferred volumes are in flight together: one output set each

    g0, r0, _ = engine.glcm_glrlm(image, mask, Ng, Nr)     # (synchronous: the dispatch verdict is read back)
    assert engine.last_path() == "sweep", "bench must run the sweep kernels, got %s" % engine.last_path()

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    cdev = device if args.backend == "nccl" else torch.device("cpu")     # where the control-plane scalars live

    def max_over_ranks(x: float) -> float:
        if not dist_on:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # Device warm-up, untimed, BEFORE the W warm-up steps and the K timed ones: the first ~20 ms of GPU work of a process run 6 - 7 %
    # slower than everything after them (scripts/r05b_coldstart.py: the same 20-step loop eight times in a fresh process gives
    # 0.523, 0.488, 0.486, 0.487 ... ms per step; W = 3 - 5 steps are 2 ms of work), and the metric is the steady rate of
    # consecutive volumes.  Every later mode of this file ran warm already; --device-warmup-ms 0 switches it off.
    # ... and the same K steps measured the way rounds 1 - 5a measured them -- right behind W warm-up steps in a process that has
    # done no other GPU work -- so that rounds stay comparable (`cold_ms_per_step` of the JSON line)
    cold_elapsed, _cf, _co = headline_loop(engine, image, mask, Ng, Nr, args.steps, args.warmup, fence, outs, families=False)
    cold_elapsed = max_over_ranks(cold_elapsed)
    prewarm_steps = 0
    if args.device_warmup_ms > 0:
        # (a fixed number of passes -- one per 10 ms asked for, 20 steps ~ 10 ms at 512^3 -- so that every rank meets the same fences)
        for _ in range(int(-(-args.device_warmup_ms // 10))):
            headline_loop(engine, image, mask, Ng, Nr, 20, 0, fence, outs, families=False)
            prewarm_steps += 20
    elapsed, fam, (glcm, glrlm) = headline_loop(engine, image, mask, Ng, Nr, args.steps, args.warmup, fence, outs)
    assert torch.equal(glcm, g0) and torch.equal(glrlm, r0), "deferred and synchronous matrices differ"
    timed_outputs = (glcm.clone(), glrlm.clone())      # what the timed pipeline left behind (the buffers are reused below)
    # the synchronous drop-in call (host waits for every volume and reads the status back), informational
    sync_steps = max(3, min(10, args.steps))
    fence()
    t1 = time.perf_counter()
    for _ in range(sync_steps):
        engine.glcm_glrlm(image, mask, Ng, Nr, out_glcm=outs[0][0], out_glrlm=outs[0][1])
    sync_ms = (time.perf_counter() - t1) / sync_steps * 1e3
    sync_fam = {f: engine.last_kernel_ms(f) for f in ("pack", "sweep", "rows", "finalize")}
    elapsed = max_over_ranks(elapsed)

    modes = None
    if not args.no_modes:
        modes = {}

        def guarded(name, fn):
            try:
                torch.cuda.empty_cache()
                modes[name] = fn()
            except Exception as e:                 # a side figure never breaks the bench line
                modes[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

        if world == 1:
            def smooth():
                im2, mk2 = make_volume(args.size, args.levels, "smooth", seed=rank, device=device)
                el, f2, _ = headline_loop(engine, im2, mk2, Ng, Nr, args.steps, args.warmup, fence, [[None, None] for _ in range(4)])
                ms2 = el / args.steps * 1e3
                return {"value": round(nvox * args.steps / el / 1e6, 1), "unit": "Mvoxels/s", "ms_per_step": round(ms2, 4),
                        "kernel": frac_of_hbm(ALG_BYTES_PER_VOXEL * nvox, f2["sweep"]), "rows_ms": round(f2["rows"], 4),
                        "job_frac": round(ALG_BYTES_PER_VOXEL * nvox / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                        "case": "the headline on low-pass filtered noise (SURVEY 8d C2(ii)): same size / levels / steps"}
            guarded("smooth", smooth)

            def levels64():
                # the headline loop at 64 grey levels: 16-bit level volume, two-table walk (kernels_sweepfw2.h)
                out = {}
                for d in ("uniform", "smooth"):
                    im2, mk2 = make_volume(args.size, 64, d, seed=rank, device=device)
                    el, f2, _ = headline_loop(engine, im2, mk2, 64, Nr, args.steps, args.warmup, fence, [[None, None] for _ in range(4)])
                    ms2 = el / args.steps * 1e3
                    out[d] = {"value": round(nvox * args.steps / el / 1e6, 1), "unit": "Mvoxels/s", "ms_per_step": round(ms2, 4),
                              "vs_32_levels": round(ms2 / (elapsed / args.steps * 1e3), 3) if d == "uniform" else None,
                              "kernel_ms": round(f2["sweep"], 4), "rows_ms": round(f2["rows"], 4),
                              "variant": engine.last_variant()}
                    del im2, mk2
                out["case"] = ("the headline loop at 64 grey levels (same size / steps), deferred calls: two-table walk with the next "
                               "volume's pack as a side job (kernel_ms), x angle on the 16-bit levels (rows_ms)")
                return out
            if args.levels == 32:
                guarded("levels64", levels64)
            guarded("config2", lambda: mode_config2(device, engine))
            guarded("config3", lambda: mode_config3(device, engine))
            guarded("fallback", lambda: mode_fallback(device, engine))
            guarded("voxel_brain1", lambda: mode_voxel_brain1(device, engine))
            guarded("voxel_wide", lambda: mode_voxel_wide(device, engine))

        def batch_mode():
            nc, dt_b, nfeat = mode_batch(device, rank, args.batch_cases, fence, world)
            dt_b = max_over_ranks(dt_b)
            out_b = {"value": round(world * nc / dt_b, 2), "unit": "cases/s", "cases_per_rank": nc, "features_per_case": nfeat,
                     "ms_per_case_per_gpu": round(dt_b / nc * 1e3, 2),
                     "one_thread_ms_per_case": round(mode_batch.one_thread_ms, 2),
                     "one_thread_loop_ms_per_case": round(mode_batch.one_thread_loop_ms, 2),
                     "case": "256^3 int16 volume from host memory, ball ROI (38 %% of the box), Original + 8 wavelet "
                             "sub-bands, six feature classes; %s" % mode_batch.how}
            if world == 1 and "PRAD_BATCH_PROCS" not in os.environ:
                # ... and the ONE-process layout beside it (what a rank gets on a host with fewer than five cores per GPU: eight
                # ranks on the 16-core bench host): three host threads of this process, executeMany
                os.environ["PRAD_BATCH_PROCS"] = "0"
                try:
                    nc1, dt1, _ = mode_batch(device, rank, args.batch_cases, fence, world)
                    out_b["one_process"] = {"value": round(nc1 / dt1, 2), "unit": "cases/s", "how": mode_batch.how}
                finally:
                    del os.environ["PRAD_BATCH_PROCS"]
            return out_b

        def voxel_mode(three_d):
            nk, dt_v, kms = mode_voxel(device, rank, world, args.size, fence, three_d)
            dt_v = max_over_ranks(dt_v)
            nk_all = nk
            if dist_on:
                t = torch.tensor([nk], dtype=torch.float64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                nk_all = int(t.item())
            # 5 B/voxel in + 8 B per centre and feature map out
            return {"value": round(nk_all / dt_v / 1e6, 2), "unit": "Mkernels/s", "kernels": nk_all,
                    "call_ms_two_runs": mode_voxel.runs_ms,      # wall clock of two consecutive calls of this rank (value: the slower)
                    "kernel_only_Mkernels_s": round(nk / (kms * 1e-3) / 1e6, 2) if kms else None,
                    "kernel": frac_of_hbm(13.0 * nk, kms),
                    "case": "%d^3 volume, GLCM JointEntropy map, %s window, every voxel a centre, centres split into "
                            "z-slabs over the ranks" % (args.size, "3-D 5x5x5 (kernelRadius 2)" if three_d else
                                                        "exampleVoxel.yaml 5x5 (force2D, kernelRadius 2)")}
        # the sharded modes run on every rank (barrier + max-over-ranks like the headline)
        try:
            modes["batch"] = batch_mode()
        except Exception as e:
            if dist_on:
                raise
            modes["batch"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        torch.cuda.empty_cache()
        modes["voxel"] = voxel_mode(False)
        torch.cuda.empty_cache()
        modes["voxel3d"] = voxel_mode(True)
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
        sweep_ms = fam["sweep"]
        alg_bytes = ALG_BYTES_PER_VOXEL * nvox
        achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9
        copy_gbps = measured_copy_bandwidth(device)
        prof = None
        try:
            with open(PROFILED_FILE) as f:
                prof = json.load(f)
            if tuple(prof.get("workload", ())) != (args.size, args.levels, args.dist) or prof.get("deferred_mode") != mode_name:
                prof = None
        except (OSError, ValueError):
            prof = None
        inline = mode_name == "pipeline" and fam["pack"] < 0.02
        out = {
            "metric": "Mvoxels/s for GLCM+GLRLM build, %d^3 vol @%d bins" % (args.size, args.levels),
            "value": round(value, 1), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "cold_ms_per_step": round(cold_elapsed / args.steps * 1e3, 4),
            "sync_call_ms_per_step": round(sync_ms, 4),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 levels / u32 counts / f64 out", "data": "synthetic",
            "modes": modes,
            "config": {"workload": "GLCM+GLRLM matrix build, %d^3 int32+uint8 volume resident in HBM, %d grey levels, "
                                   "full mask, 13 angles, %s levels; one volume per GPU (batch sharding, no collective)"
                                   % (args.size, args.levels, args.dist),
                       "size": args.size, "levels": args.levels, "dist": args.dist, "deferred_mode": mode_name,
                       "device_warmup": {"untimed_steps_before_the_W_warmup_steps": prewarm_steps, "ms": args.device_warmup_ms,
                                         "why": "the first ~20 ms of GPU work of a process run 6 - 7 % slower than the steady state "
                                                "(profiles/r05b_probes.md section 6); the timed region is the K steps after the W warm-up steps"}},
            "roofline": {
                "bound": "hbm",
                "kernel": "sweep_fw_kernel: the 12 line angles of one volume" +
                          (" + the pack (5 B/voxel read, 1 B/voxel written) of the next volume as a side job of the same "
                           "launch" if inline else "") + "; the x angle is sweep_fw_rows_kernel (rows_ms)",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": prof["bytes"] if prof else None,
                "traffic_kernel": prof["kernel_bytes"] if prof else None,
                "traffic_source": prof["source"] if prof else "rocprofv3 --pmc cannot run inside bench.py; see profiles/",
                "algorithmic_bytes": alg_bytes, "kernel_ms": round(sweep_ms, 4), "rows_ms": round(fam["rows"], 4),
                "pack_ms": round(fam["pack"], 4), "finalize_ms": round(fam["finalize"], 4),
                "pipeline_ms": round(fam["device"], 4),
                "pipeline_achieved": round(alg_bytes / (fam["device"] * 1e-3) / 1e9, 2),
                "measured_copy_GBps": round(copy_gbps, 1), "frac_of_measured_copy": round(achieved / copy_gbps, 5),
                "job_achieved": round(alg_bytes / (ms * 1e-3) / 1e9, 2),
                "job_frac": round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                "sync_call_kernel_ms": {k: round(v, 4) for k, v in sync_fam.items()},
                "note": "achieved = 5 B/voxel x voxels / duration of the dominant launch, HIP events on the launch stream "
                        "inside the timed region, two records per step (all launches of a step sit on one stream in pipeline "
                        "mode, so a launch's duration is the kernel's); rows_ms / pack_ms / finalize_ms / pipeline_* (the whole "
                        "call: walks + x angle + finalize + memsets between its events) come from a fully instrumented pass of "
                        "the same loop after the timed region -- ten event records per step cost the stream 6 - 12 %; "
                        "job_* = 5 B/voxel over ms_per_step (wall, max over ranks)",
                "kernel_ms_instrumented_pass": round(fam.get("sweep_instrumented", 0.0), 4),
            },
        }
        if prof and "lds" in prof:
            t = sweep_ms * 1e-3
            out["roofline"]["secondary"] = {
                "bound": "lds_atomic", "achieved": round(prof["lds"] / t / 1e9, 2), "peak": round(LDS_ATOMIC_PEAK / 1e9, 2),
                "unit": "G ds_add wave-instructions/s", "frac": round(prof["lds"] / t / LDS_ATOMIC_PEAK, 4),
                "note": "one LDS atomic per run end; each ds_add holds the LDS operand bus 4.1 cycles per CU",
                "source": prof["source"]}
            out["roofline"]["secondary_valu"] = {
                "bound": "valu", "achieved": round(prof["valu"] / t / 1e9, 2), "peak": round(VALU_PEAK / 1e9, 2),
                "unit": "G VALU wave-instructions/s", "frac": round(prof["valu"] / t / VALU_PEAK, 4),
                "source": prof["source"]}
        if world == 1 and not args.no_host_boundary:
            try:
                out["host_boundary"] = host_boundary(image, mask, Ng, Nr)
            except Exception as e:
                out["host_boundary"] = {"error": str(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_full(engine, image, mask, Ng, Nr, args, timed_outputs)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
