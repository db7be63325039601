"""pretrain_gp with SOD on the drop-in classes (the bench line's `pretrain` rows on their own): python tools/time_pretrain.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mcp_boot  # noqa: F401
from mc_pilco_amd import workloads

dev = torch.device("cuda", 0)
for shape in ("cartpole", "ur5"):
    r = workloads.time_pretrain(dev, shape)
    print(shape, json.dumps({k: (v if k != "stage_us" else {a: round(b, 1) for a, b in v.items()}) for k, v in r.items()}))
