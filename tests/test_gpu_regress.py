"""Regression fixtures of mismatches the randomised stress (scripts/r05_stress.py) found in round 5 -- both in the fixed-window
walk of rounds 2-4, both invisible to the earlier tests (they need marches of more than 64 steps, i.e. pieces, and runs of a
particular length):
  1. a DEAD line (a piece that begins inside a run) of level 1 whose age the checked path had clamped read as "safe, length 0"
     in margin<false>, walked on on the plain path, grew into the state (level 1, length 1) and recorded its next change as a
     short run of level 1;
  2. a run of exactly RS + 1 voxels that ended at the x edge of a window-filling row on the plain path landed in the slot of
     "longer than RS", which only the GLCM reads.
Fixture: tests/golden/regress/fw_long_runs_138x58x300.npz (constant slabs along y with 2 % noise: runs of up to 138 voxels
along z and the diagonals), the failing case and the crops / piece lengths that separated the two."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regress", "fw_long_runs_138x58x300.npz")


@pytest.mark.parametrize("crop,env", [((138, 58, 300), {}), ((138, 58, 300), {"PRAD_FW_CL": "32"}), ((138, 58, 300), {"PRAD_FW_RS": "16"}),
                                      ((138, 58, 256), {}), ((138, 58, 256), {"PRAD_FW_CL": "144"}), ((138, 58, 256), {"PRAD_FW_RS": "16"}),
                                      ((128, 44, 256), {}), ((138, 8, 300), {})])
def test_long_runs_across_pieces_and_at_the_row_edge(crop, env, checker, monkeypatch):
    from pyradiomics_amd import cmatrices as cm, _lib
    d = np.load(FIX)
    img = np.ascontiguousarray(d["img"].astype(np.int32)[:crop[0], :crop[1], :crop[2]])
    Ng = int(d["Ng"])
    tiled = np.concatenate([img, img[:, :, :512 - crop[2]]], axis=2) if crop[2] == 256 and not env else None
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for vol in [img] + ([tiled] if tiled is not None else []):
        msk = np.ones(vol.shape, bool)
        Nr = max(vol.shape)
        g, r, ang = cm.calculate_glcm_glrlm(vol, msk, Ng, Nr, False, 0)
        assert _lib.last_path() == "sweep" and _lib.last_variant() == "fw"
        wg, wang = checker.calculate_glcm(vol, msk, [1], Ng, False, 0)
        wr, _ = checker.calculate_glrlm(vol, msk, Ng, Nr, False, 0)
        assert np.array_equal(ang, wang)
        assert np.array_equal(r, wr), "GLRLM %s %s" % (vol.shape, env)
        assert np.array_equal(g, wg), "GLCM %s %s" % (vol.shape, env)


def test_same_fixture_through_the_deferred_pipeline(checker):
    """the launch bench.py times (PACK = true: volume N packed by the launch that walks volume N - 1) on the 512-wide tiling of
    the fixture: every volume in flight bit-exact"""
    import torch
    from bench import headline_loop
    from pyradiomics_amd import engine
    d = np.load(FIX)
    img = d["img"].astype(np.int32)[:, :, :256]
    vol = np.ascontiguousarray(np.concatenate([img, img], axis=2))
    Ng, Nr = int(d["Ng"]), 512
    msk = np.ones(vol.shape, bool)
    wg, _ = checker.calculate_glcm(vol, msk, [1], Ng, False, 0)
    wr, _ = checker.calculate_glrlm(vol, msk, Ng, Nr, False, 0)
    dev = torch.device("cuda", 0)
    img_d, msk_d = torch.from_numpy(vol).to(dev), torch.from_numpy(msk.astype(np.uint8)).to(dev)
    outs = [[None, None] for _ in range(4)]
    headline_loop(engine, img_d, msk_d, Ng, Nr, 4, 2, torch.cuda.synchronize, outs, families=False)
    assert engine.last_path() == "sweep"
    for o in outs:
        assert np.array_equal(o[0].cpu().numpy(), wg.reshape(o[0].shape)), "GLCM (deferred pipeline)"
        assert np.array_equal(o[1].cpu().numpy(), wr.reshape(o[1].shape)), "GLRLM (deferred pipeline)"
