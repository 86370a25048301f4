#!/usr/bin/env python
"""Generator of tests/golden/notebook_brain1.json: the outputs the REFERENCE ITSELF recorded when its maintainers ran
/root/reference/notebooks/helloFeatureClass.ipynb (executed cells are stored in the notebook).  They pin the filter
stack to what PyWavelets / SimpleITK really compute:

  * cell "logFeatures" (ipynb:1487-1559): `imageoperations.getLoGImage(image, mask, sigma=[1.0, 3.0, 5.0])` on the whole
    brain1 image, `cropToTumorMask(..., bb)`, `RadiomicsFirstOrder(**inputSettings).enableAllFeatures().execute()`
    -> 3 x 18 values `log-sigma-<s>-0-mm-3D_<feature>`
  * cell "waveletFeatures" (ipynb:1609-1772): the same with `getWaveletImage(image, mask)` (coif1, level 1)
    -> 8 x 18 values `wavelet-<band>_<feature>`
  * the cells on the unfiltered crop (first order Mean, GLCM, GLRLM, GLSZM with settings binWidth 25) are kept as well
    under "original".

The generator callbacks pass NO settings to the feature class (getLoGImage / getWaveletImage yield their own kwargs,
which hold no binWidth), so the first-order class runs on its defaults: binWidth 25, voxelArrayShift 0.

Run in the dev container (reads /root/reference, which does not exist on the GPU box); the JSON it writes is committed.
    python tests/golden/make_notebook_golden.py"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
NOTEBOOK = "/root/reference/notebooks/helloFeatureClass.ipynb"
LINE = re.compile(r"^\s*([A-Za-z0-9_.\-]+)\s*:\s*([-+0-9.eEnaif]+)\s*$")


def cell_text(cell):
    out = []
    for o in cell.get("outputs", []):
        if "text" in o:
            out.append("".join(o["text"]))
    return "".join(out)


def parse(text):
    vals = {}
    for line in text.splitlines():
        m = LINE.match(line)
        if m:
            vals[m.group(1)] = float(m.group(2))
    return vals


def main():
    nb = json.load(open(NOTEBOOK))
    golden = {"source": "notebooks/helloFeatureClass.ipynb (executed outputs stored by the reference)",
              "case": "brain1", "settings": {"binWidth": 25}, "log": {}, "wavelet": {}, "original": {}}
    cells = nb["cells"]
    for i, c in enumerate(cells):
        if c["cell_type"] != "code":
            continue
        src = "".join(c["source"])
        vals = parse(cell_text(c))
        if not vals:
            continue
        if "laplacianFeatureName" in src:
            for k, v in vals.items():
                image_type, feature = k.rsplit("_", 1)
                golden["log"].setdefault(image_type, {})[feature] = v
        elif "waveletFeatureName" in src:
            for k, v in vals.items():
                image_type, feature = k.rsplit("_", 1)
                golden["wavelet"].setdefault(image_type, {})[feature] = v
        else:
            for cls in ("firstOrder", "glcm", "glrlm", "glszm"):
                if "result = %sFeatures.execute()" % cls in src:
                    golden["original"][cls.lower()] = vals
    assert sorted(golden["log"]) == ["log-sigma-1-0-mm-3D", "log-sigma-3-0-mm-3D", "log-sigma-5-0-mm-3D"], golden["log"].keys()
    assert len(golden["wavelet"]) == 8 and all(len(v) == 18 for v in golden["wavelet"].values())
    assert all(len(v) == 18 for v in golden["log"].values())
    path = os.path.join(HERE, "notebook_brain1.json")
    with open(path, "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)
    n = sum(len(v) for v in golden["log"].values()) + sum(len(v) for v in golden["wavelet"].values())
    print("wrote %s: %d filter values, original: %s" % (path, n, {k: len(v) for k, v in golden["original"].items()}))


if __name__ == "__main__":
    main()
