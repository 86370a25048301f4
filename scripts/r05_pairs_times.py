"""round 5: device ms of the calls the sweeps do not take (bench.py modes.fallback on its own)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # noqa: E402
from pyradiomics_amd import engine             # noqa: E402

print(json.dumps(bench.mode_fallback(torch.device("cuda", 0), engine), indent=1))
