"""GPU parity at BASELINE.json's own sizes, against the reference's own cmatrices.c when
oracle/_ref/libcmatrices_ref.so travelled with the repo (else our C restatement):
  C2  synthetic 256^3 volume, full mask, 32 grey levels, ALL FIVE matrices, iid and smooth levels;
  C4  voxel-based GLCM JointEntropy with the exampleVoxel.yaml window (force2D, kernelRadius 2) on a 512^3 volume:
      >= 10^4 sampled kernel centres, fused device feature vs the reference's per-kernel matrix + the numpy formula
      of glcm.py:560-576;
  HEADLINE  the bench volume itself, 512^3 at 32 levels, uniform and smooth: GLCM + GLRLM bit for bit against the
      reference C run angle by angle over the host's cores (cmatrices.c:4-92, :299-541 take the angle table as an
      argument), for the synchronous call AND for what the deferred pipeline of bench.py's timed loop leaves behind."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _volume(n, kind, seed):
    import torch
    from bench import make_volume
    img, msk = make_volume(n, 32, kind, seed, torch.device("cuda", 0))
    return img, msk


def test_execute_many_overlaps_cases_and_equals_execute():
    """RadiomicsFeatureExtractor.executeMany (case i + 1's head before case i's last image is collected, one thread) and
    batch.run_batch(many=, threads=2): every value equals execute()'s bit for bit, in the order of the cases; a failing case
    re-raises at its position and the cases behind it still run on a fresh call"""
    import torch
    from pyradiomics_amd import batch
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    from pyradiomics_amd.image import Image
    rng = np.random.default_rng(5)
    N = 56
    zz, yy, xx = np.ogrid[:N, :N, :N]
    roi = (((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.42 * N) ** 2).astype(np.int16)
    from scipy import ndimage
    vols = [(ndimage.gaussian_filter(rng.standard_normal((N, N, N)), 1.2) * 400 + 800).astype(np.int16) for _ in range(6)]
    ex = RadiomicsFeatureExtractor({"setting": {"binCount": 24, "additionalInfo": False},
                                    "imageType": {"Original": {}, "Wavelet": {}, "LoG": {"sigma": [1.5]}}})
    seq = [ex.execute(Image(v), Image(roi)) for v in vols]
    many = list(ex.executeMany((Image(v), Image(roi)) for v in vols))
    thr = batch.run_batch(vols, None, threads=2, many=lambda cs: ex.executeMany((Image(v), Image(roi)) for v in cs))
    torch.cuda.synchronize()
    assert len(seq) == len(many) == len(thr) == 6 and len(seq[0]) > 900
    for a, b, c in zip(seq, many, thr):
        assert list(a) == list(b) == list(c)
        for k in a:
            x, y, z = float(a[k]), float(b[k]), float(c[k])
            assert (x == y == z) or (np.isnan(x) and np.isnan(y) and np.isnan(z)), k
    # a case without its label in the middle: the cases in front of it arrive, the error surfaces, nothing is left in flight
    empty = np.zeros_like(roi)
    got = []
    with pytest.raises(ValueError):
        for r in ex.executeMany([(Image(vols[0]), Image(roi)), (Image(vols[1]), Image(empty)), (Image(vols[2]), Image(roi))]):
            got.append(r)
    assert len(got) <= 1
    again = list(ex.executeMany((Image(v), Image(roi)) for v in vols[:3]))
    for a, b in zip(seq[:3], again):
        assert all(float(a[k]) == float(b[k]) or np.isnan(float(a[k])) for k in a)


@pytest.mark.parametrize("kind", ["uniform", "smooth"])
def test_c2_all_five_matrices_256(kind, checker):
    from pyradiomics_amd import cmatrices as cm, _lib
    n, Ng = 256, 32
    img_d, msk_d = _volume(n, kind, 3)
    img, msk = img_d.cpu().numpy(), msk_d.cpu().numpy().astype(bool)
    g, r, ang = cm.calculate_glcm_glrlm(img, msk, Ng, n, False, 0)
    assert _lib.last_path() == "sweep"
    eg, eang = checker.calculate_glcm(img, msk, [1], Ng, False, 0)
    assert np.array_equal(ang, eang) and np.array_equal(g, eg), "GLCM"
    er, _ = checker.calculate_glrlm(img, msk, Ng, n, False, 0)
    assert np.array_equal(r, er), "GLRLM"
    assert np.array_equal(cm.calculate_gldm(img, msk, [1], Ng, 0, False, 0), checker.calculate_gldm(img, msk, [1], Ng, 0, False, 0)), "GLDM"
    a, b = cm.calculate_ngtdm(img, msk, [1], Ng, False, 0), checker.calculate_ngtdm(img, msk, [1], Ng, False, 0)
    assert np.array_equal(a[..., 0], b[..., 0]) and np.array_equal(a[..., 2], b[..., 2]), "NGTDM counts"
    np.testing.assert_allclose(a[..., 1], b[..., 1], rtol=1e-10, atol=0)   # float column: summation order (1e-6 is the bar)
    Ns = int(msk.sum())
    assert np.array_equal(cm.calculate_glszm(img, msk, Ng, Ns, False, 0), checker.calculate_glszm(img, msk, Ng, Ns, False, 0)), "GLSZM"


def test_c4_voxel_joint_entropy_512_sampled(checker):
    import torch
    from pyradiomics_amd import engine
    n, Ng, nk = 512, 32, 12000
    img_d, msk_d = _volume(n, "smooth", 5)
    rng = np.random.default_rng(11)
    vox = np.stack([rng.integers(0, n, nk), rng.integers(0, n, nk), rng.integers(0, n, nk)]).astype(np.int32)
    vox[:, :64] = np.array([[0, 0, 0], [n - 1, n - 1, n - 1], [0, n - 1, 0], [n // 2, 0, n - 1]], np.int32).T.repeat(16, 1)  # corners
    kw = dict(kernelRadius=2, force2D=True, force2Ddimension=0)
    got = engine.voxel_glcm_features(img_d, msk_d, Ng, torch.from_numpy(vox).cuda(), ["JointEntropy"], **kw)["JointEntropy"]
    assert engine.last_path() == "voxel-fused"
    got = got.cpu().numpy()
    # the reference's route: per-kernel matrices from the C checker, then glcm.py:145-188 (symmetrise, drop empty
    # angles, normalise) and :560-576 (JointEntropy = -sum p log2(p + eps), mean over angles)
    img, msk = img_d.cpu().numpy(), msk_d.cpu().numpy().astype(bool)
    want = np.empty(nk)
    for s in range(0, nk, 2000):
        P, _ = checker.calculate_glcm(img, msk, [1], Ng, True, 0, kernelRadius=2, voxels=vox[:, s:s + 2000])
        P = P + P.transpose(0, 2, 1, 3)
        tot = P.sum((1, 2))
        with np.errstate(invalid="ignore", divide="ignore"):
            p = P / tot[:, None, None, :]
            ent = -(p * np.log2(p + np.spacing(1))).sum((1, 2))
        ent[tot == 0] = np.nan
        want[s:s + 2000] = np.nanmean(ent, axis=1)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12)


WIDE_FEATS = ["Autocorrelation", "ClusterProminence", "ClusterShade", "ClusterTendency", "Contrast", "DifferenceAverage",
              "DifferenceVariance", "Id", "Idm", "Idn", "Idmn", "InverseVariance", "SumAverage", "SumSquares"]


def _reference_voxel_glcm(checker, img, msk, Ng, vox, force2D, chunk=1000, wide=False):
    """The reference's voxel-based route for the features the sliding-window kernel carries: per-kernel matrices from
    the C checker (_cmatrices.c:203-222, set_bb :1120-1147), then glcm.py:149-205 (symmetrise, empty angles -- none of the
    dense call's angles is empty -- NaN marks, normalise), :226-258 (ux, HXY) and the feature formulas (:260-887): JointAverage =
    plain mean over the angles, everything else nanmean.  `wide`: also the fourteen features of the kernel's WIDE instantiation
    (the sums over p_{x-y}(k) and p_{x+y}(k) written as weighted sums over p(i, j): the same numbers in another summation order)."""
    import warnings
    nk = vox.shape[1]
    names = ["JointEntropy", "JointEnergy", "JointAverage"] + (WIDE_FEATS if wide else [])
    out = {f: np.empty(nk) for f in names}
    lev = np.arange(1, Ng + 1, dtype=float)
    I, J = lev[None, :, None, None], lev[None, None, :, None]
    K = np.abs(I - J)
    seen = None
    for s in range(0, nk, chunk):
        P, _ = checker.calculate_glcm(img, msk, [1], Ng, force2D, 0, kernelRadius=2, voxels=np.ascontiguousarray(vox[:, s:s + chunk]))
        P = P + P.transpose(0, 2, 1, 3)
        tot = P.sum((1, 2))
        seen = (tot > 0).any(0) if seen is None else seen | (tot > 0).any(0)
        tot[tot == 0] = np.nan
        sl = slice(s, s + chunk)
        with np.errstate(invalid="ignore", divide="ignore"), warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            p = P / tot[:, None, None, :]
            out["JointEntropy"][sl] = np.nanmean(-(p * np.log2(p + np.spacing(1))).sum((1, 2)), 1)
            out["JointEnergy"][sl] = np.nanmean((p ** 2).sum((1, 2)), 1)
            ux = (I * p).sum((1, 2), keepdims=True)
            out["JointAverage"][sl] = ux[:, 0, 0, :].mean(1)
            if wide:
                def nm(w):
                    return np.nanmean((p * w).sum((1, 2)), 1)
                c = I + J - 2 * ux                                                    # (symmetric matrix: uy = ux)
                da = (p * K).sum((1, 2), keepdims=True)
                out["Autocorrelation"][sl] = nm(I * J)
                out["ClusterProminence"][sl] = nm(c ** 4)
                out["ClusterShade"][sl] = nm(c ** 3)
                out["ClusterTendency"][sl] = nm(c ** 2)
                out["Contrast"][sl] = nm((I - J) ** 2)
                out["DifferenceAverage"][sl] = np.nanmean(da[:, 0, 0, :], 1)
                out["DifferenceVariance"][sl] = nm((K - da) ** 2)
                out["Id"][sl] = nm(1.0 / (1.0 + K))
                out["Idm"][sl] = nm(1.0 / (1.0 + K ** 2))
                out["Idn"][sl] = nm(1.0 / (1.0 + K / Ng))
                out["Idmn"][sl] = nm(1.0 / (1.0 + K ** 2 / Ng ** 2))
                out["InverseVariance"][sl] = nm(np.where(K > 0, 1.0 / np.where(K > 0, K, 1.0) ** 2, 0.0))
                out["SumAverage"][sl] = nm(I + J)
                out["SumSquares"][sl] = nm((I - ux) ** 2)
    assert seen.all(), "an angle without a single pair in the sample: the empty-angle rule of glcm.py:186-199 would apply"
    return out


def _slab_sample(rng, z0, z1, n, count):
    """`count` centres of the slab z0 <= z < z1: its corners, edges and faces first, random ones after"""
    zs, ys = [z0, z0 + 1, z1 - 2, z1 - 1], [0, 1, 2, n // 2, n - 3, n - 2, n - 1]
    fixed = np.array([(z, y, x) for z in zs for y in ys for x in ys], np.int32).T
    r = np.stack([rng.integers(z0, z1, count - fixed.shape[1]), rng.integers(0, n, count - fixed.shape[1]),
                  rng.integers(0, n, count - fixed.shape[1])]).astype(np.int32)
    return np.concatenate([fixed, r], 1)


@pytest.mark.parametrize("three_d", [False, True])
def test_c4_sliding_window_maps_512_dense_vs_reference(three_d, checker):
    """VERDICT r4 missing #1 / weak #1: the kernel config 4's figure is quoted on, voxel_glcm_slide_kernel, against the
    REFERENCE route (not against this repo's window kernel).  Every voxel of two z-slabs of the 512^3 bench volume is a centre
    (dense requests: the route the bench takes, asserted), exampleVoxel.yaml's 5 x 5 window and the 3-D 5^3 window, full mask
    and a partial mask (random holes, an empty band, ROI-less centres); >= 10^4 of the returned centres per window -- slab
    corners, edges, faces, centres whose neighbours are masked out -- are compared with the reference's per-kernel matrices
    + numpy formulas at 1e-9."""
    import torch
    from pyradiomics_amd import engine
    n, Ng = 512, 32
    img_d, msk_d = _volume(n, "smooth", 0)                   # the volume of bench.py's mode_voxel
    g = torch.Generator(device=img_d.device)
    g.manual_seed(17)
    part_d = (torch.rand((n, n, n), generator=g, device=img_d.device) < 0.8).to(torch.uint8)
    part_d[:, 200:212, :] = 0                                # windows without a single ROI voxel
    part_d[:, :, 300:303] = 0
    img = img_d.cpu().numpy()
    feats = ["JointEntropy", "JointEnergy", "JointAverage"]
    rng = np.random.default_rng(23)
    kw = dict(kernelRadius=2, force2D=not three_d, force2Ddimension=0)
    compared = 0
    for mask_d in (msk_d, part_d):
        msk = mask_d.cpu().numpy().astype(bool)
        for z0, z1 in ((0, 16), (500, 512)):
            zz, yy, xx = torch.meshgrid(torch.arange(z0, z1, device=img_d.device, dtype=torch.int32),
                                        torch.arange(n, device=img_d.device, dtype=torch.int32),
                                        torch.arange(n, device=img_d.device, dtype=torch.int32), indexing="ij")
            vox_d = torch.stack([zz.reshape(-1), yy.reshape(-1), xx.reshape(-1)])
            del zz, yy, xx
            got = engine.voxel_glcm_features(img_d, mask_d, Ng, vox_d, feats, **kw)
            assert engine.last_path() == "voxel-fused" and engine.last_variant() == "slide"
            pick = _slab_sample(rng, z0, z1, n, 2600)
            flat = torch.from_numpy(((pick[0] - z0).astype(np.int64) * n + pick[1]) * n + pick[2]).to(img_d.device)
            want = _reference_voxel_glcm(checker, img, msk, Ng, pick, not three_d)
            for f in feats:
                a, b = got[f][flat].cpu().numpy(), want[f]
                assert np.array_equal(np.isnan(a), np.isnan(b)), (f, z0, int(np.isnan(a).sum()), int(np.isnan(b).sum()))
                ok = ~np.isnan(b)
                np.testing.assert_allclose(a[ok], b[ok], rtol=1e-9, atol=1e-12, err_msg="%s slab %d" % (f, z0))
            if z0 == 0:          # the WIDE instantiation (fourteen more features) on the same slab and sample
                gotw = engine.voxel_glcm_features(img_d, mask_d, Ng, vox_d, WIDE_FEATS, **kw)
                assert engine.last_variant() == "slide"
                sub = pick[:, :900]
                wantw = _reference_voxel_glcm(checker, img, msk, Ng, sub, not three_d, wide=True)
                for f in WIDE_FEATS:
                    a, b = gotw[f][flat[:900]].cpu().numpy(), wantw[f]
                    assert np.array_equal(np.isnan(a), np.isnan(b)), (f, z0)
                    ok = ~np.isnan(b)
                    np.testing.assert_allclose(a[ok], b[ok], rtol=1e-9, atol=1e-11, err_msg="%s slab %d" % (f, z0))
                del gotw
            if mask_d is part_d:
                assert np.isnan(want["JointAverage"]).sum() > 20 and (~msk[pick[0], pick[1], pick[2]]).sum() > 100
            compared += pick.shape[1]
            del got, vox_d
    assert compared >= 10000


def test_c4_example_voxel_yaml_on_brain1_takes_the_sliding_window_kernel(checker):
    """VERDICT r4 missing #4: the reference's OWN example -- examples/exampleSettings/exampleVoxel.yaml (binWidth 25, force2D,
    kernelRadius 2, maskedKernel, GLCM JointEntropy) on data/brain1 -- has 33 grey levels, one more than round 4's sliding-window
    kernel took.  The crop the reference makes (bounding box padded by kernelRadius, featureextractor.py:385-395), numpy
    binning, every ROI voxel a centre: the request must take the sliding-window kernel and every value must equal the
    reference route (per-kernel matrices of the reference C + glcm.py's numpy formulas) at 1e-9."""
    import os
    import torch
    from helpers import GOLDEN
    from pyradiomics_amd import engine, imageoperations
    from pyradiomics_amd.image import read_nrrd
    image = read_nrrd(os.path.join(GOLDEN, "data", "brain1_image.nrrd"))
    mask = read_nrrd(os.path.join(GOLDEN, "data", "brain1_label.nrrd"))
    ci, cmk = imageoperations.cropToTumorMask(image, mask, 1, padDistance=2)
    roi = cmk.array == 1
    levels, _ = imageoperations.binImage(ci.array, roi, binWidth=25)
    levels = np.where(roi, levels, 0).astype(np.int32)
    Ng = int(levels.max())
    assert Ng == 33 and int(roi.sum()) == 4137
    vox = np.array(np.nonzero(roi)).astype(np.int32)
    dev = torch.device("cuda", 0)
    feats = ["JointEntropy", "JointEnergy", "JointAverage"]
    got = engine.voxel_glcm_features(torch.from_numpy(levels).to(dev), torch.from_numpy(roi.astype(np.uint8)).to(dev), Ng,
                                     torch.from_numpy(vox).to(dev), feats, kernelRadius=2, force2D=True, force2Ddimension=0)
    assert engine.last_path() == "voxel-fused" and engine.last_variant() == "slide"
    want = _reference_voxel_glcm(checker, levels, roi, Ng, vox, True, wide=True)
    for f in feats:
        a, b = got[f].cpu().numpy(), want[f]
        assert np.array_equal(np.isnan(a), np.isnan(b)), f
        ok = ~np.isnan(b)
        np.testing.assert_allclose(a[ok], b[ok], rtol=1e-9, atol=1e-12, err_msg=f)
    # the fourteen features of the kernel's WIDE instantiation, alone and together with the three above
    for req in (WIDE_FEATS, feats + WIDE_FEATS, ["Contrast"], ["Idmn", "ClusterProminence"]):
        gw = engine.voxel_glcm_features(torch.from_numpy(levels).to(dev), torch.from_numpy(roi.astype(np.uint8)).to(dev), Ng,
                                        torch.from_numpy(vox).to(dev), req, kernelRadius=2, force2D=True, force2Ddimension=0)
        assert engine.last_variant() == "slide", req
        for f in req:
            a, b = gw[f].cpu().numpy(), want[f]
            assert np.array_equal(np.isnan(a), np.isnan(b)), f
            ok = ~np.isnan(b)
            np.testing.assert_allclose(a[ok], b[ok], rtol=1e-9, atol=1e-11, err_msg=f)


def test_c3_filters_rebinning_matrices_256_full_size(checker):
    """VERDICT r4 missing #2: BASELINE config 3 at its own size -- the 256^3 bench volume through the product's wavelet
    (8 coif1 sub-bands) and LoG (sigma 1..5 mm) kernels against oracle/filters_oracle.py (pinned by the reference's notebook),
    then, on the PRODUCT's filtered images (matrix parity on shared inputs), binCount-32 levels against
    numpy's getBinEdges / binImage arithmetic (imageoperations.py:119-126,156-174) and GLCM + GLRLM of each of the 13 level
    volumes bit for bit against the reference C (cmatrices.c:4-92, :299-541), one (matrix, angle) per host thread."""
    import torch
    from bench import make_volume
    from oracle import filters_oracle as fo
    from pyradiomics_amd import engine
    n = 256
    dev = torch.device("cuda", 0)
    lv, msk_d = make_volume(n, 32, "smooth", 0, dev)
    img_d = (lv.to(torch.float32) * 25.0 + 3.0).to(torch.int16)          # bench.py mode_config3's volume
    img = img_d.cpu().numpy()
    msk = msk_d.cpu().numpy().astype(bool)
    derived = dict(engine.wavelet_images(img_d))
    assert len(derived) == 8
    ap, ret = fo.swt3(img, "coif1")
    want = {"wavelet-" + k: v for k, v in ret[0].items()}
    want["wavelet-LLL"] = ap
    del ap, ret
    assert set(want) == set(derived)
    for name in sorted(want):
        w = want.pop(name)
        g = derived[name].cpu().numpy()
        assert g.dtype == np.float64 and np.abs(g - w).max() <= 1e-9 * np.abs(w).max(), name
        del w, g
    sig = (1.0, 2.0, 3.0, 4.0, 5.0)
    for s, d in zip(sig, engine.log_images(img_d, (1.0, 1.0, 1.0), sig)):
        w = fo.laplacian_recursive_gaussian(img, (1.0, 1.0, 1.0), s)
        g = d.cpu().numpy()
        assert g.dtype == np.float32 and np.abs(g - w).max() <= 2e-6 * np.abs(w).max(), s
        derived["log-sigma-%g" % s] = d
        del w, g
    assert len(derived) == 13
    for name, d in derived.items():
        levels, Ng, edges = engine.bin_image(d, msk_d, binCount=32)[:3]
        x = d.cpu().numpy()
        e = np.histogram(x[msk], 32)[1]                                  # imageoperations.py:122-126
        e[-1] += 1
        assert np.array_equal(np.asarray(edges, dtype=np.float64), e), name
        lev = np.digitize(x, e).astype(np.int32)                         # :156-174 (full mask: every voxel is binned)
        assert Ng == int(lev.max()) == 32
        assert np.array_equal(levels.cpu().numpy(), lev), name
        g, r, ang = engine.glcm_glrlm(levels, msk_d, Ng, n)
        assert engine.last_path() == "sweep"
        want_g, want_r, want_ang, _ = checker.glcm_glrlm_angle_sharded(lev, msk, Ng, n)
        assert np.array_equal(ang, want_ang)
        assert np.array_equal(g.cpu().numpy(), want_g), "GLCM " + name
        assert np.array_equal(r.cpu().numpy(), want_r), "GLRLM " + name


def test_c5_case_256_all_classes_vs_reference_route(checker):
    """VERDICT r4 missing #2: ONE case of BASELINE config 5 at its own size (bench.py's batch case: 256^3 int16 volume, ball
    ROI, Original + 8 wavelet sub-bands, six feature classes, binCount 32) through RadiomicsFeatureExtractor.execute on the
    HIP backend -- device-resident, fused formulas, the route bench.py times -- against the reference-shaped route assembled
    from the checkers: oracle wavelet (filters_oracle.swt3 on the whole image, featureextractor.py:371-395), crop to the ROI
    box, numpy binning, the reference C matrices (oracle/_ref) + the classes' numpy formulas, numpy first-order statistics.
    Every one of the 837 feature values at 1e-6 relative."""
    import torch
    from bench import _batch_case_setup
    from helpers import feature_class, FEATURE_CLASSES
    from oracle import filters_oracle as fo
    from pyradiomics_amd import backend, cmatrices
    from pyradiomics_amd.image import Image
    ex, vols, one = _batch_case_setup(torch.device("cuda", 0), 4000, 1)
    backend.set(cmatrices)
    got = one(0)
    keys = [k for k in got if not k.startswith("diagnostics")]
    assert len(keys) == 9 * (18 + 24 + 16 + 16 + 14 + 5)
    vol = vols[0]
    N = vol.shape[0]
    zz, yy, xx = np.ogrid[:N, :N, :N]
    roi = (((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.45 * N) ** 2)
    idx = np.nonzero(roi)
    bb = tuple(slice(int(i.min()), int(i.max()) + 1) for i in idx)
    ap, ret = fo.swt3(vol, "coif1")
    derived = {"original": vol}
    derived.update({"wavelet-" + k: v for k, v in ret[0].items()})
    derived["wavelet-LLL"] = ap
    mask_c = Image(np.ascontiguousarray(roi[bb]).astype(np.int16))
    backend.set(checker)
    worst = (0.0, None)
    try:
        for name, arr in derived.items():
            img_c = Image(np.ascontiguousarray(arr[bb]))
            for cls in FEATURE_CLASSES:
                fc = feature_class(cls)(img_c, mask_c, binCount=32, deviceResident=False)
                fc.enableAllFeatures()
                for fname, ref in fc.execute().items():
                    k = "%s_%s_%s" % (name, cls, fname)
                    if k not in got:
                        continue                      # (deprecated features are not in the extractor's default set)
                    a, b = float(got[k]), float(ref)
                    if np.isnan(b):
                        assert np.isnan(a), k
                        continue
                    err = abs(a - b) / max(abs(b), 1e-300)
                    assert err <= 1e-6, (k, a, b, err)
                    if err > worst[0]:
                        worst = (err, k)
                    keys.remove(k)
    finally:
        backend.set(cmatrices)
    assert keys == [], keys[:5]
    print("config 5 case: worst relative error %.3g (%s)" % worst)


def test_cases_in_flight_on_threads_equal_the_sequential_run():
    """batch.run_batch(threads=3): three whole cases at a time on one GPU (per-thread library contexts and HIP streams);
    every feature value equals the one-at-a-time run bit for bit"""
    import torch
    from pyradiomics_amd import batch
    from pyradiomics_amd.featureextractor import RadiomicsFeatureExtractor
    from pyradiomics_amd.image import Image
    rng = np.random.default_rng(3)
    N = 48
    zz, yy, xx = np.ogrid[:N, :N, :N]
    roi = (((zz - N / 2) ** 2 + (yy - N / 2) ** 2 + (xx - N / 2) ** 2) < (0.42 * N) ** 2).astype(np.int16)
    from scipy import ndimage
    vols = [(ndimage.gaussian_filter(rng.standard_normal((N, N, N)), 1.5) * 400 + 800).astype(np.int16) for _ in range(7)]
    ex = RadiomicsFeatureExtractor({"setting": {"binCount": 16, "additionalInfo": False},
                                    "imageType": {"Original": {}, "Wavelet": {}}})

    def one(v):
        return ex.execute(Image(v), Image(roi))

    seq = batch.run_batch(vols, one)
    par = batch.run_batch(vols, one, threads=3)
    torch.cuda.synchronize()
    assert len(seq) == len(par) == 7
    for a, b in zip(seq, par):
        assert list(a) == list(b)
        for k in a:
            x, y = float(a[k]), float(b[k])
            assert x == y or (np.isnan(x) and np.isnan(y)), k


@pytest.mark.parametrize("kind", ["uniform", "smooth"])
def test_headline_512_bit_exact(kind, checker):
    """VERDICT r3 'missing #2': the headline configuration at FULL size, not a slab and not only identities"""
    import torch
    from bench import headline_loop
    from pyradiomics_amd import engine
    n, Ng = 512, 32
    img_d, msk_d = _volume(n, kind, 0)                       # seed 0 = rank 0's volume in bench.py
    want_g, want_r, want_ang, info = checker.glcm_glrlm_angle_sharded(img_d.cpu().numpy(), msk_d.cpu().numpy().astype(bool), Ng, n)
    g, r, ang = engine.glcm_glrlm(img_d, msk_d, Ng, n)       # synchronous
    assert engine.last_path() == "sweep" and engine.last_variant().startswith("fw")
    assert np.array_equal(ang, want_ang)
    assert np.array_equal(g.cpu().numpy(), want_g), "GLCM (synchronous call)"
    assert np.array_equal(r.cpu().numpy(), want_r), "GLRLM (synchronous call)"
    # the timed loop of bench.py: deferred steps back to back, the pack of volume N riding in the walk launch of N-1
    outs = [[None, None] for _ in range(4)]
    _, _, (gd, rd) = headline_loop(engine, img_d, msk_d, Ng, n, 4, 2, torch.cuda.synchronize, outs)
    assert np.array_equal(gd.cpu().numpy(), want_g), "GLCM (deferred pipeline)"
    assert np.array_equal(rd.cpu().numpy(), want_r), "GLRLM (deferred pipeline)"
    for o in outs:                                            # every volume in flight, not only the last
        assert np.array_equal(o[0].cpu().numpy(), want_g) and np.array_equal(o[1].cpu().numpy(), want_r)
