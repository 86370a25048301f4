"""Host logic of the case pipeline that needs no GPU: how cmatrices.segment_image_enqueue reads the result block of
prad_image_enqueue_dev (layout in include/pyradiomics_amd.h), with a stand-in for the engine call."""
import types

import numpy as np
import pytest


class _FakeTensor:
    """enough of a tensor for the code under test (it only passes them through)"""
    shape = (4, 5, 6)


def _install_fake_engine(monkeypatch, block, layout):
    from pyradiomics_amd import cmatrices
    calls = {}

    fake = types.SimpleNamespace(
        IMG_GLCM=1, IMG_GLRLM=2, IMG_GLDM=4, IMG_NGTDM=8, IMG_GLSZM=16, IMG_FIRSTORDER=32, IMG_MCC=64,
        FIRSTORDER_FIELDS=("Np", "Energy", "Minimum", "P10", "P25", "Median", "P75", "P90", "Maximum", "Mean", "MAD", "rMAD",
                           "m2", "m3", "m4"))

    def image_enqueue(levels, mask, raw, Ng, Ns, classes, **kw):
        calls["classes"] = classes
        calls["kw"] = kw
        return {"res": block, "layout": layout, "ticket": 0, "keep": None}

    fake.image_enqueue = image_enqueue
    fake.glszm_compact = lambda *a, **k: calls.setdefault("three_calls", True) and (np.ones((3, 2)), np.array([1, 2]))
    fake.zone_matrix_features = lambda P, sizes: (np.arange(16, dtype=float)[None] + 100.0, np.array([False]))
    fake.firstorder_stats = lambda *a, **k: {"Np": -1.0}
    import pyradiomics_amd
    monkeypatch.setattr(pyradiomics_amd, "engine", fake, raising=False)
    monkeypatch.setitem(__import__("sys").modules, "pyradiomics_amd.engine", fake)
    monkeypatch.setattr(cmatrices, "_to_device", lambda a, **k: a)
    return calls


def test_result_block_is_read_by_its_layout(monkeypatch):
    from pyradiomics_amd import cmatrices
    Na = 3
    block = np.full(400, np.nan)
    layout = [-1] * 16
    layout[11] = Na
    # GLCM: 3 angles x 23 values, flags right behind them; angle 1 is empty and must not enter the mean
    layout[0], layout[1] = 0, Na * 23
    vals = np.arange(Na * 23, dtype=float).reshape(Na, 23)
    vals[1] = np.nan
    block[:Na * 23] = vals.ravel()
    flags = block[layout[1]:layout[1] + 3].view(np.int32)
    flags[:Na] = [0, 1, 0]
    # MCC: per-angle values + verdict 0
    layout[2] = 80
    block[80:84] = [0.5, np.nan, 0.7, 0.0]
    # NGTDM
    layout[7] = 96
    block[96:101] = [1, 2, 3, 4, 5]
    # GLSZM: verdict != 0 -> the three-call route
    layout[8], layout[9] = 104, 128
    block[104:121] = list(range(16)) + [2.0]
    block[128:129].view(np.int32)[0] = 0
    # first order: 15 statistics + verdict 0
    layout[10] = 136
    block[136:152] = list(np.arange(15) + 0.25) + [0.0]
    calls = _install_fake_engine(monkeypatch, block, layout)
    reqs = {"glcm": {"features": ["Contrast", "MCC"], "symmetrical": False},
            "ngtdm": {"features": ["Busyness", "Strength"]},
            "glszm": {"features": ["ZoneEntropy"]},
            "glrlm": {"features": ["NotAFeature"]},                  # outside the fused table: left to the caller
            "firstorder": {"raw": _FakeTensor(), "shift": 3.0, "features": None}}
    tok, fin = cmatrices.segment_image_enqueue(_FakeTensor(), _FakeTensor(), 12, 1000, reqs)
    assert tok["ticket"] == 0 and set(fin) == {"glcm", "ngtdm", "glszm", "firstorder"}
    assert calls["classes"] == 1 | 64 | 8 | 16 | 32 and calls["kw"]["symmetric"] is False and calls["kw"]["voxelArrayShift"] == 3.0
    g = fin["glcm"]()
    k = cmatrices.VOXEL_GLCM_FEATURES.index("Contrast")
    assert g["Contrast"] == pytest.approx((vals[0, k] + vals[2, k]) / 2) and g["MCC"] == pytest.approx(0.6)
    n = fin["ngtdm"]()
    names = cmatrices._ZONE_LIKE["ngtdm"][1]
    assert n == {"Busyness": block[96 + names.index("Busyness")], "Strength": block[96 + names.index("Strength")]}
    z = fin["glszm"]()                                                  # verdict 2: recomputed by the exact route
    assert calls.get("three_calls") and z["ZoneEntropy"] == 100.0 + cmatrices._ZONE_LIKE["glszm"][1].index("ZoneEntropy")
    f = fin["firstorder"]()
    assert f["Np"] == 0.25 and f["m4"] == 14.25


def test_declined_parts_and_the_mcc_verdict(monkeypatch):
    from pyradiomics_amd import cmatrices
    Na = 2
    block = np.zeros(200)
    layout = [-1] * 16
    layout[11] = Na
    layout[0], layout[1] = 0, Na * 23
    layout[2] = 56
    block[56:59] = [0.1, 0.2, 1.0]          # verdict: more grey levels than the device MCC takes
    layout[10] = 64
    block[64:80] = [0.0] * 15 + [8.0]        # first-order queue declined at run time
    _install_fake_engine(monkeypatch, block, layout)
    reqs = {"glcm": {"features": ["MCC", "Idm"], "symmetrical": True}, "gldm": {"features": ["DependenceEntropy"], "alpha": 0},
            "firstorder": {"raw": _FakeTensor(), "shift": 0.0, "features": None}}
    tok, fin = cmatrices.segment_image_enqueue(_FakeTensor(), _FakeTensor(), 80, 5, reqs)
    assert set(fin) == {"glcm", "firstorder"}            # GLDM's part came back as -1: the caller queues it on its own
    g = fin["glcm"]()
    assert "MCC" not in g and "Idm" in g                  # the class then takes its host route for MCC
    assert fin["firstorder"]() == {"Np": -1.0}            # the synchronous statistics
