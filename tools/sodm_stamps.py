"""Diagnostic (SODM_STAMPS build: python -c "import build; build.build_variant('sodm', ['SODM_STAMPS'], only=['gp_pretrain.hip'])" in mc-pilco_amd/):
cycles per accepted point of sod_select_multi_kernel's intervals, wave 0 and wave 8 of the middle workgroup.
    MCPILCO_HIP_EXPERIMENT=1 MCPILCO_HIP_LIB=mc-pilco_amd/libmcpilco_hip_sodm.so python tools/sodm_stamps.py [N] [thr]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, numpy as np, torch, ctypes as C
from mc_pilco_amd import hipabi as abi, ops, workloads

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dev = torch.device("cuda", 0)
pb = workloads.numpy_problem("ur5_script", N=N)
spec = workloads.spec_for(pb["cfg"], pb["cfg"]["sigma_n"], None if pb["poly"] is None else pb["poly"][0])
X = torch.tensor(pb["Z"], dtype=torch.float64, device=dev)
nbytes = abi.lib().mcp_sod_workspace_bytes(N)
ws = torch.zeros((nbytes + 7) // 8, dtype=torch.float64, device=dev)
idx = torch.zeros(N, dtype=torch.int32, device=dev)
n = torch.zeros(1, dtype=torch.int32, device=dev)
kc = spec.to_c(dev)
for _ in range(2):
    abi.check(abi.lib().mcp_sod_select(C.byref(kc), N, abi.ptr(X), thr, abi.ptr(idx), abi.ptr(n), abi.ptr(ws), nbytes, abi.stream()), "sod")
torch.cuda.synchronize()
cnt = int(n.item())
st = ws[N * N:N * N + 16].view(torch.int64).cpu().tolist()
names = ["dot (to barrier 1)", "barrier 1", "wave 0: reduce, test, publish index", "barrier 2", "publish vector", "indices / predicted vector", "barrier 3", "pivot + miss path + barrier 4"]
print("N %d kept %d" % (N, cnt))
for w, off in (("wave 0", 0), ("wave 8", 8)):
    print(w, " | ".join("%s %.0f" % (nm, st[off + k] / max(cnt, 1)) for k, nm in enumerate(names)), "| total %.0f cycles per point" % (sum(st[off:off + 8]) / max(cnt, 1)))
