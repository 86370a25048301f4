"""GPU parity of the fixed-window sweep kernel (csrc/kernels_sweepfw.h: rows of 65..512 voxels, fused GLCM+GLRLM table)
against the CPU checkers: the reference's own cmatrices.c (oracle/_ref) when that prebuilt file travelled with the
repo, otherwise our C restatement.  Bit-exact, every edge of the design exercised: rows that fill the window exactly
(256 with 4 columns per lane, 512 with 8) and ragged ones, pieces + tails (PRAD_FW_CL forces many short pieces),
rows that wrap (dy != 0 with Ny smaller / larger than Nz), runs longer than the table's length slots, flat volumes,
partial masks, a mask with whole empty planes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cm():
    from pyradiomics_amd import cmatrices
    return cmatrices


def _levels(seed, shape, Ng, kind):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.integers(1, Ng + 1, size=shape).astype(np.int32)
    if kind == "flat":
        return np.full(shape, 1 + seed % Ng, np.int32)
    from scipy import ndimage
    sigma = 2.0 if kind == "smooth" else 6.0        # "blobs": runs of tens of voxels
    f = ndimage.gaussian_filter(rng.standard_normal(shape), sigma)
    f = (f - f.min()) / (f.max() - f.min() + 1e-12)
    return np.minimum((f * Ng).astype(np.int32) + 1, Ng)


def _mask(seed, shape, kind):
    rng = np.random.default_rng(seed + 99)
    if kind == "full":
        return np.ones(shape, bool)
    if kind == "random":
        return rng.random(shape) < 0.7
    zz, yy, xx = np.meshgrid(*[np.linspace(-1, 1, n) for n in shape], indexing="ij")
    return (zz ** 2 + yy ** 2 + xx ** 2) < 0.8        # ball: empty corners, long outside stretches


def _check(cm, checker, img, mask, Ng):
    from pyradiomics_amd import _lib
    Nr = int(max(img.shape))
    g, r, ang = cm.calculate_glcm_glrlm(img, mask, Ng, Nr, False, 0)
    assert _lib.last_path() == "sweep"
    eg, eang = checker.calculate_glcm(img, mask, [1], Ng, False, 0)
    er, _ = checker.calculate_glrlm(img, mask, Ng, Nr, False, 0)
    assert np.array_equal(ang, eang)
    bad = np.argwhere(g != eg)
    assert bad.size == 0, "GLCM differs at %d entries, first (i, j, angle) = %s: %s vs %s; angle %s" % (
        len(bad), bad[0], g[tuple(bad[0])], eg[tuple(bad[0])], ang[bad[0][-1]])
    bad = np.argwhere(r != er)
    assert bad.size == 0, "GLRLM differs at %d entries, first (i, len-1, angle) = %s: %s vs %s; angle %s" % (
        len(bad), bad[0], r[tuple(bad[0])], er[tuple(bad[0])], ang[bad[0][-1]])


SHAPES = [(20, 24, 512), (24, 20, 256), (18, 30, 511), (30, 18, 257), (9, 40, 300), (40, 9, 65), (26, 26, 128),
          (33, 12, 200), (12, 33, 500), (70, 16, 72)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("kind", ["uniform", "smooth"])
def test_fw_shapes(cm, checker, shape, kind):
    img = _levels(hash(shape) % 997, shape, 32, kind)
    _check(cm, checker, img, _mask(1, shape, "full"), 32)


@pytest.mark.parametrize("shape", [(20, 24, 512), (30, 18, 257), (24, 20, 256)])
@pytest.mark.parametrize("mkind", ["random", "ball"])
@pytest.mark.parametrize("kind", ["uniform", "blobs"])
def test_fw_masks(cm, checker, shape, mkind, kind):
    img = _levels(5, shape, 32, kind)
    _check(cm, checker, img, _mask(2, shape, mkind), 32)


@pytest.mark.parametrize("shape", [(40, 44, 512), (64, 20, 256), (24, 70, 130)])
@pytest.mark.parametrize("kind", ["blobs", "flat"])
def test_fw_long_runs(cm, checker, shape, kind):
    img = _levels(3, shape, 16, kind)
    _check(cm, checker, img, _mask(3, shape, "full"), 16)
    if kind == "flat":   # a flat slab inside noise: every line crosses it, runs end on its faces at the same step
        img2 = _levels(4, shape, 16, "uniform")
        img2[shape[0] // 4: shape[0] // 2, 3:-3, 10:-10] = 7
        _check(cm, checker, img2, _mask(3, shape, "full"), 16)


@pytest.mark.parametrize("cl", ["8", "16", "24"])
@pytest.mark.parametrize("shape", [(37, 29, 512), (50, 21, 256), (23, 41, 100)])
def test_fw_short_pieces(cm, checker, shape, cl, monkeypatch):
    """every piece boundary + tail combination: pieces of 8..24 steps over walks of 21..50 steps"""
    monkeypatch.setenv("PRAD_FW_CL", cl)
    for kind, mkind in (("uniform", "full"), ("blobs", "ball"), ("smooth", "random")):
        _check(cm, checker, _levels(8, shape, 24, kind), _mask(4, shape, mkind), 24)


@pytest.mark.parametrize("Ng", [2, 7, 40])
def test_fw_levels(cm, checker, Ng):
    shape = (21, 22, 320)
    _check(cm, checker, _levels(6, shape, Ng, "uniform"), _mask(5, shape, "random"), Ng)
    _check(cm, checker, _levels(6, shape, Ng, "smooth"), _mask(5, shape, "full"), Ng)


def test_fw_empty_planes_and_rows(cm, checker):
    shape = (30, 30, 512)
    img = _levels(9, shape, 32, "uniform")
    mask = np.ones(shape, bool)
    mask[10:14] = False
    mask[:, 5:9] = False
    mask[:, :, 100:140] = False
    _check(cm, checker, img, mask, 32)


def test_fw_matches_wrapped_lines_kernel(cm, monkeypatch):
    """the two line kernels against each other on a volume the CPU checkers would take minutes for"""
    shape = (160, 192, 512)
    img = _levels(12, shape, 32, "smooth")
    mask = _mask(7, shape, "ball")
    g1, r1, _ = cm.calculate_glcm_glrlm(img, mask, 32, 512, False, 0)
    monkeypatch.setenv("PRAD_NO_FW", "1")
    g0, r0, _ = cm.calculate_glcm_glrlm(img, mask, 32, 512, False, 0)
    assert np.array_equal(g0, g1) and np.array_equal(r0, r1)


@pytest.mark.parametrize("shape", [(10, 12, 1024), (12, 10, 600), (8, 14, 513), (14, 8, 1000), (20, 9, 768)])
def test_fw_rows_of_513_to_1024_voxels(cm, checker, shape, monkeypatch):
    """rows beyond one 512-column window: 16 columns per lane (K = 16); exact fill (1024), ragged, masks, long runs, pieces"""
    from pyradiomics_amd import _lib
    for kind, mkind in (("uniform", "full"), ("smooth", "ball"), ("blobs", "random"), ("flat", "full")):
        _check(cm, checker, _levels(hash(shape) % 983, shape, 32, kind), _mask(5, shape, mkind), 32)
        assert _lib.last_variant() == "fw"
    monkeypatch.setenv("PRAD_FW_CL", "8")
    _check(cm, checker, _levels(3, shape, 32, "smooth"), _mask(6, shape, "ball"), 32)


@pytest.mark.parametrize("shape", [(24, 30, 512), (300, 12, 128), (12, 300, 72), (128, 128, 128)])
@pytest.mark.parametrize("level", [11, 32])
def test_fw_very_long_runs_of_high_levels(cm, checker, shape, level):
    """a flat volume of a HIGH level: every run spans its whole line (up to 512 voxels).  Until round 3 a dead line was one
    whose state lay behind the table, and level*P + len*Q of such a run crossed that mark -- its end was dropped (the x
    angle of a (24, 30, 512) volume of level 11 lost GLRLM[10][511] and the GLCM diagonal that derives from it)"""
    img = np.full(shape, level, np.int32)
    img[1, 2, 3] = 5                               # (and one voxel that cuts a few of the runs)
    _check(cm, checker, img, _mask(1, shape, "full"), 32)
    _check(cm, checker, img, _mask(3, shape, "ball"), 32)


@pytest.fixture(params=["pipeline", "lanes"])
def deferred_mode(request):
    """both ways deferred calls overlap: the two-stage pipeline (volume N's pack rides in the walk launch of volume N-1)
    and the lanes (internal streams)"""
    from pyradiomics_amd import engine
    engine.set_deferred_mode(1 if request.param == "pipeline" else 0)
    yield request.param
    engine.set_deferred_mode(-1)


def test_deferred_calls_pipeline_and_report_late(cm, deferred_mode):
    """deferred mode: calls only enqueue; results equal the synchronous call; irregular levels surface in the status"""
    import torch
    from pyradiomics_amd import engine
    shape, Ng = (40, 48, 512), 32
    vols = [(_levels(s, shape, Ng, "uniform"), _mask(s, shape, "random")) for s in (1, 2, 3)]
    dev = [(torch.from_numpy(i).cuda(), torch.from_numpy(m.astype(np.uint8)).cuda()) for i, m in vols]
    want = [engine.glcm_glrlm(i, m, Ng, 512) for i, m in dev]
    want = [(g.clone(), r.clone()) for g, r, _ in want]
    got = [engine.glcm_glrlm(i, m, Ng, 512, deferred=True) for i, m in dev]     # three volumes in flight
    engine.deferred_status()
    for (g, r, _), (eg, er) in zip(got, want):
        assert torch.equal(g, eg) and torch.equal(r, er)
    bad = vols[0][0].copy()
    bad[3, 4, 5] = Ng + 1
    mk = np.ones(shape, np.uint8)
    engine.glcm_glrlm(torch.from_numpy(bad).cuda(), torch.from_numpy(mk).cuda(), Ng, 512, deferred=True)
    with pytest.raises(RuntimeError):
        engine.deferred_status()
    engine.deferred_status()       # the flag was cleared by the query


@pytest.mark.parametrize("lanes", [1, 2, 3])
def test_deferred_lanes_agree(cm, lanes, deferred_mode):
    """deferred whole-volume calls alternate between the library's lanes (internal streams + workspaces): five volumes
    of different shapes in flight, every result equal to the synchronous call; inputs produced on the caller's stream
    right before the call (the lane has to wait for them)"""
    import torch
    from pyradiomics_amd import engine
    Ng = 24
    shapes = [(30, 40, 512), (44, 28, 256), (30, 40, 512), (25, 33, 300), (44, 28, 256)]
    vols = [(_levels(10 + i, s, Ng, "smooth" if i % 2 else "uniform"), _mask(20 + i, s, "ball" if i % 2 else "full"))
            for i, s in enumerate(shapes)]
    want = []
    for i, m in vols:
        g, r, _ = engine.glcm_glrlm(torch.from_numpy(i).cuda(), torch.from_numpy(m.astype(np.uint8)).cuda(), Ng, 512)
        want.append((g.clone(), r.clone()))
    engine.set_lanes(lanes)
    try:
        got = []
        for i, m in vols:
            di = torch.from_numpy(i).cuda(non_blocking=True) + 0      # (a kernel on the caller's stream feeds the call)
            dm = torch.from_numpy(m.astype(np.uint8)).cuda(non_blocking=True)
            got.append(engine.glcm_glrlm(di, dm, Ng, 512, deferred=True))
        engine.deferred_status()
        for (g, r, _), (eg, er) in zip(got, want):
            assert torch.equal(g, eg) and torch.equal(r, er)
    finally:
        engine.set_lanes(0)
    with pytest.raises(ValueError):
        engine.set_lanes(9)


def test_lanes_survive_a_workspace_release_and_mixed_sizes(cm, deferred_mode):
    """deferred volumes of alternating shapes (every call re-plans and regrows its lane's workspace), a workspace
    release between two deferred batches, a synchronous call in between: all results equal the synchronous ones"""
    import torch
    from pyradiomics_amd import engine
    Ng = 16
    shapes = [(20, 24, 512), (33, 21, 128), (20, 24, 512), (12, 50, 300), (40, 40, 64), (33, 21, 128)]
    vols = [(torch.from_numpy(_levels(30 + i, s, Ng, "blobs")).cuda(), torch.from_numpy(_mask(40 + i, s, "random").astype(np.uint8)).cuda())
            for i, s in enumerate(shapes)]
    want = []
    for i, m in vols:
        g, r, _ = engine.glcm_glrlm(i, m, Ng, 512)
        want.append((g.clone(), r.clone()))
    for round_ in range(2):
        got = [engine.glcm_glrlm(i, m, Ng, 512, deferred=True) for i, m in vols[:3]]
        g_sync, r_sync, _ = engine.glcm_glrlm(vols[4][0], vols[4][1], Ng, 512)        # a synchronous call in between
        got += [engine.glcm_glrlm(i, m, Ng, 512, deferred=True) for i, m in vols[3:]]
        engine.deferred_status()
        assert torch.equal(g_sync, want[4][0]) and torch.equal(r_sync, want[4][1])
        for (g, r, _), (eg, er) in zip(got, want):
            assert torch.equal(g, eg) and torch.equal(r, er)
        engine.release_workspace()


def test_deferred_join_orders_the_callers_stream(cm, deferred_mode):
    """deferred_join(): the caller's stream waits for the lanes on the device; torch work queued afterwards sees the
    outputs without any host synchronisation in between"""
    import torch
    from pyradiomics_amd import engine
    Ng, shape = 16, (40, 48, 512)
    vols = [(torch.from_numpy(_levels(50 + i, shape, Ng, "uniform")).cuda(), torch.from_numpy(_mask(60 + i, shape, "full").astype(np.uint8)).cuda())
            for i in range(4)]
    want = [engine.glcm_glrlm(i, m, Ng, 512)[0].sum().item() for i, m in vols]
    got = [engine.glcm_glrlm(i, m, Ng, 512, deferred=True) for i, m in vols]
    engine.deferred_join()
    sums = torch.stack([g.sum() for g, _, _ in got])          # on the caller's stream, right behind the join
    engine.deferred_status()
    assert sums.tolist() == want


def test_pipeline_packs_inside_the_previous_walk_and_flushes(cm, checker):
    """pipeline mode in detail: the pack of volume N is a side job of volume N-1's walk launch (no pack launch of its own
    from the second call on: the "pack" timing family stays empty), a volume whose rows are no multiple of 16 voxels or
    that is no fixed-window volume takes its own pack / the ordinary route, partial masks raise row flags through the side
    job, and the last volume is walked by the flush; every result bit-equal to the CPU checker"""
    import torch
    from pyradiomics_amd import engine
    Ng = 32
    shapes = [(24, 30, 512), (24, 30, 512), (20, 26, 256), (18, 22, 300), (24, 30, 512), (30, 30, 64), (24, 30, 512)]
    kinds = ["uniform", "smooth", "uniform", "blobs", "flat", "uniform", "smooth"]
    masks = ["full", "ball", "random", "full", "full", "random", "ball"]
    vols = [(_levels(70 + i, s, Ng, k), _mask(80 + i, s, m)) for i, (s, k, m) in enumerate(zip(shapes, kinds, masks))]
    dev = [(torch.from_numpy(i).cuda(), torch.from_numpy(m.astype(np.uint8)).cuda()) for i, m in vols]
    engine.set_deferred_mode(1)
    try:
        engine.timing_begin()
        got = [engine.glcm_glrlm(i, m, Ng, 512, deferred=True) for i, m in dev]
        engine.deferred_status()
        pack_ms = engine.timing_ms("pack")
        engine.timing_end()
        for n, ((img, mask), (g, r, _)) in enumerate(zip(vols, got)):
            eg, _ = checker.calculate_glcm(img, mask, [1], Ng, False, 0)
            er, _ = checker.calculate_glrlm(img, mask, Ng, 512, False, 0)
            assert np.array_equal(g.cpu().numpy(), eg[0]), "GLCM of volume %d %s" % (n, shapes[n])
            assert np.array_equal(r.cpu().numpy(), er[0]), "GLRLM of volume %d %s" % (n, shapes[n])
        # same-shape volumes back to back: only the first one packs in a launch of its own
        engine.timing_begin()
        got = [engine.glcm_glrlm(dev[0][0], dev[0][1], Ng, 512, deferred=True) for _ in range(4)]
        engine.deferred_status()
        t_pack, t_sweep = engine.timing_ms("pack"), engine.timing_ms("sweep")
        engine.timing_end()
        assert t_sweep > 0 and pack_ms > 0
        one = engine.glcm_glrlm(dev[0][0], dev[0][1], Ng, 512)
        engine.timing_begin()
        engine.glcm_glrlm(dev[0][0], dev[0][1], Ng, 512, deferred=True)
        engine.deferred_status()
        t_one_pack = engine.timing_ms("pack")
        engine.timing_end()
        assert t_pack < 1.5 * t_one_pack + 0.01, (t_pack, t_one_pack)     # 4 volumes, ONE standalone pack
        for g, r, _ in got:
            assert torch.equal(g, one[0]) and torch.equal(r, one[1])
    finally:
        engine.set_deferred_mode(-1)


@pytest.mark.parametrize("odd", [0, 33, 300, -1, 70000])
def test_side_job_pack_handles_odd_level_words(cm, checker, odd):
    """the pack that rides in the previous volume's walk converts four voxels at a time with packed-byte arithmetic; whatever
    is not the plain case takes the exact per-voxel form: levels with bits beyond the low byte OUTSIDE the mask (ignored,
    like the reference: cmatrices.c:61-64 tests the mask first), and an irregular level UNDER the mask (the call reports
    it, the synchronous route raises the reference's IndexError)"""
    import torch
    from pyradiomics_amd import engine, _lib
    shape, Ng = (24, 30, 512), 32
    engine.set_deferred_mode(1)
    try:
        base = _levels(5, shape, Ng, "uniform")
        mask = _mask(6, shape, "random")
        junk = base.copy()
        rng = np.random.default_rng(7)
        out = ~mask.astype(bool)
        junk[out] = rng.choice(np.array([0, -5, 255, 256, 1 << 20, -(1 << 30), 33, 44], dtype=np.int32), size=int(out.sum()))
        vols = [(base, mask), (junk, mask), (base, mask)]
        dev = [(torch.from_numpy(i).cuda(), torch.from_numpy(m.astype(np.uint8)).cuda()) for i, m in vols]
        got = [engine.glcm_glrlm(i, m, Ng, 512, deferred=True) for i, m in dev]      # volumes 2 and 3 pack as side jobs
        engine.deferred_status()
        eg, _ = checker.calculate_glcm(base, mask, [1], Ng, False, 0)
        er, _ = checker.calculate_glrlm(base, mask, Ng, 512, False, 0)
        for g, r, _ in got:                                     # junk outside the mask changes nothing
            assert np.array_equal(g.cpu().numpy(), eg[0]) and np.array_equal(r.cpu().numpy(), er[0])
        bad = base.copy()
        zz, yy, xx = np.nonzero(mask)
        k = len(zz) // 2
        bad[zz[k], yy[k], xx[k]] = odd                          # one irregular voxel under the mask
        devbad = torch.from_numpy(bad).cuda()
        engine.glcm_glrlm(dev[0][0], dev[0][1], Ng, 512, deferred=True)
        engine.glcm_glrlm(devbad, dev[0][1], Ng, 512, deferred=True)               # packed by the side job
        engine.glcm_glrlm(dev[0][0], dev[0][1], Ng, 512, deferred=True)
        with pytest.raises(_lib.DeferredLevelsError):
            engine.deferred_status()
        engine.deferred_status()
        try:
            want = checker.calculate_glcm(bad, mask, [1], Ng, False, 0)
        except IndexError:
            with pytest.raises(IndexError):
                cm.calculate_glcm(bad, mask, [1], Ng, False, 0)
        else:                                                   # (the reference aliases some out-of-range levels silently)
            assert np.array_equal(cm.calculate_glcm(bad, mask, [1], Ng, False, 0)[0], want[0])
    finally:
        engine.set_deferred_mode(-1)


# ---- the two-table fixed-window kernel (csrc/kernels_sweepfw2.h): 45+ grey levels ------------------------------------------
def _check2(cm, checker, img, mask, Ng):
    from pyradiomics_amd import _lib
    _check(cm, checker, img, mask, Ng)
    assert _lib.last_variant() == "fw2", _lib.last_variant()


@pytest.mark.parametrize("Ng", [44, 64, 100, 128, 160])
@pytest.mark.parametrize("shape", [(20, 24, 512), (24, 20, 256), (18, 30, 300), (30, 18, 130), (9, 40, 66)])
def test_fw2_levels_and_shapes(cm, checker, Ng, shape):
    """Ng 45 .. 160 on the fixed-window kernel with two tables and 16-bit level elements: rows that fill the window (256
    with 4 columns per lane, 512 with 8) and ragged ones, iid and smooth levels, bit-exact against the reference C"""
    for kind in ("uniform", "smooth"):
        _check2(cm, checker, _levels(hash(shape) % 991 + Ng, shape, Ng, kind), _mask(1, shape, "full"), Ng)


@pytest.mark.parametrize("Ng", [64, 128])
@pytest.mark.parametrize("mkind", ["random", "ball"])
def test_fw2_masks(cm, checker, Ng, mkind):
    for shape, kind in (((20, 24, 512), "uniform"), ((30, 18, 257), "blobs"), ((24, 20, 256), "smooth")):
        _check2(cm, checker, _levels(5 + Ng, shape, Ng, kind), _mask(2, shape, mkind), Ng)


@pytest.mark.parametrize("Ng", [64, 150])
def test_fw2_long_runs_pieces_and_flat_volumes(cm, checker, Ng, monkeypatch):
    """runs longer than the table's length slots (Ng = 150 leaves ~100 of them; PRAD_FW2_RS forces 24), flat volumes of a
    high level, many short pieces (dead lines + tails)"""
    for shape in ((40, 44, 512), (64, 20, 256), (24, 70, 130)):
        for kind in ("blobs", "flat"):
            img = _levels(9 + Ng, shape, Ng, kind)
            if kind == "flat":
                img[:] = Ng
                img[1, 2, 3] = 5
            _check2(cm, checker, img, _mask(3, shape, "full"), Ng)
    monkeypatch.setenv("PRAD_FW2_RS", "24")
    monkeypatch.setenv("PRAD_FW_CL", "16")
    for shape in ((40, 44, 512), (70, 16, 200)):
        for kind in ("blobs", "smooth", "flat"):
            _check2(cm, checker, _levels(11 + Ng, shape, Ng, kind), _mask(4, shape, "ball" if kind != "flat" else "full"), Ng)


def test_fw2_equals_the_lines_kernels_and_serves_deferred_calls(cm, monkeypatch):
    import torch
    from pyradiomics_amd import engine
    Ng, shape = 64, (48, 40, 512)
    img = torch.from_numpy(_levels(77, shape, Ng, "smooth")).cuda()
    mask = torch.from_numpy(_mask(7, shape, "ball").astype(np.uint8)).cuda()
    g1, r1, _ = engine.glcm_glrlm(img, mask, Ng, 512)
    assert engine.last_variant() == "fw2"
    got = [engine.glcm_glrlm(img, mask, Ng, 512, deferred=True) for _ in range(3)]
    engine.deferred_status()
    for g, r, _ in got:
        assert torch.equal(g, g1) and torch.equal(r, r1)
    monkeypatch.setenv("PRAD_NO_FW2", "1")
    g0, r0, _ = engine.glcm_glrlm(img, mask, Ng, 512)
    assert engine.last_variant() == "lines"
    assert torch.equal(g0, g1) and torch.equal(r0, r1)


@pytest.mark.parametrize("Ng", [64, 100])
def test_fw2_runs_of_length_one_are_derived_exactly(cm, checker, Ng, monkeypatch):
    """the two-table walk does not record runs of length 1 when the x angle is in the call: the finalize step restores
    GLRLM_a[g][1] = N_g - sum_{len >= 2} len * GLRLM_a[g][len] from the x angle's runs (kernels_sweepfw2.h, SKIP1).  Cases that
    lean on it: iid levels (31 of 32 runs have length 1), sparse masks whose voxels are isolated along some angles (the
    reference's rule cmatrices.c:524-534 then clears that angle's length-1 column AFTER the restore), a call without the x
    angle (force2D across x: every run is recorded), and the recording path forced by PRAD_FW2_NOSKIP1"""
    from pyradiomics_amd import _lib
    shape = (22, 26, 512)
    img = _levels(123 + Ng, shape, Ng, "uniform")
    rng = np.random.default_rng(5)
    sparse = rng.random(shape) < 0.004           # nearly every ROI voxel isolated
    three = np.zeros(shape, bool)
    three[3, 4, 5] = three[3, 4, 6] = three[10, 20, 300] = True      # one pair along x, one lone voxel
    for mask in (_mask(1, shape, "full"), _mask(8, shape, "random"), sparse, three):
        _check2(cm, checker, img, mask, Ng)
    Nr = int(max(shape))
    for dim in (0, 1, 2):                        # in-plane angle lists; dim 2 drops every angle with an x component
        g, r, ang = cm.calculate_glcm_glrlm(img, sparse | _mask(2, shape, "ball"), Ng, Nr, True, dim)
        assert _lib.last_path() == "sweep"
        eg, eang = checker.calculate_glcm(img, sparse | _mask(2, shape, "ball"), [1], Ng, True, dim)
        er, _ = checker.calculate_glrlm(img, sparse | _mask(2, shape, "ball"), Ng, Nr, True, dim)
        assert np.array_equal(ang, eang) and np.array_equal(g, eg) and np.array_equal(r, er), dim
    monkeypatch.setenv("PRAD_FW2_NOSKIP1", "1")
    _check2(cm, checker, img, _mask(8, shape, "random"), Ng)
    _check2(cm, checker, _levels(7, shape, Ng, "smooth"), sparse, Ng)
