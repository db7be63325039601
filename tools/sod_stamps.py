"""Per-section cycle totals of sod_select_kernel (SOD_STAMPS build: python mc-pilco_amd/build.py --variant-gp sodst SOD_STAMPS;
MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=.../libmcpilco_hip_sodst.so python tools/sod_stamps.py): waves 0 (dead early), 4 (group 0, live to the end), 4 of group 1."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import mcp_boot  # noqa: F401
from mc_pilco_amd import hipabi as abi, ops, synthetic as sy

dev = torch.device("cuda", 0)
c = sy.CARTPOLE
Z, Ys = sy.gp_io(sy.cartpole_rollouts(n_roll=5), c["angle"], c["not_angle"], c["vel"])
N = 300
X = torch.as_tensor(Z[:N]).to(dev).contiguous()
spec = ops.KernelSpec(torch.as_tensor(c["lengthscales"], dtype=torch.float64), 1.0, 0.36 ** 2)
nb = abi.lib().mcp_sod_workspace_bytes(N)
ws = torch.zeros((nb + 7) // 8, dtype=torch.float64, device=dev)
ix = torch.zeros(N, dtype=torch.int32, device=dev)
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
kc = spec.to_c(dev)
for _ in range(3):
    abi.check(abi.lib().mcp_sod_select(C.byref(kc), N, abi.ptr(X), 0.18, abi.ptr(ix), abi.ptr(cnt), abi.ptr(ws), nb, abi.stream()), "sod")
torch.cuda.synchronize()
st = ws[(N - 1) * N:(N - 1) * N + 24].view(torch.int64).cpu().numpy().reshape(3, 8)
names = ["top+gather+kern", "barrier 1", "dot", "barrier 2 (partials)", "sum+finalize", "barrier 3"]
for w, row in zip(("wave 0", "wave 4 grp 0", "wave 4 grp 1"), st):
    n = max(int(row[6]), 1)
    print(w, "accepted", n, " per accepted point:", ", ".join("%s %.0f" % (a, b / n) for a, b in zip(names, row[:6])), " total %.0f cycles" % (row[:6].sum() / n))
