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
