"""Diagnostic (experiment build with CLX_STAMPS only): where a block row of the left-looking Cholesky spends its cycles."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcp_boot, numpy as np, torch
from mc_pilco_amd import hipabi, ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.RandomState(0)
A = rs.randn(N, N + 3)
K = torch.tensor(A @ A.T / (N + 3) + 0.1 * np.eye(N), dtype=torch.float64, device="cuda:0")
for _ in range(3):
    ops.chol_factor(K)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (16 * 128))()
L = ctypes.CDLL(hipabi.LIB_PATH)
L.mcp_debug_read_chol_stamps(buf)
st = np.array(buf, dtype=np.int64).reshape(128, 16)
nb = (N + 15) // 16
print("row: w0 [P->cols, factor, inverse, wait b1, phase e+b2] | w1 [mirror stores, sums, wait b1 + e] | row total")
for I in range(nb):
    r = st[I]
    nxt = st[I + 1][0] if I + 1 < nb else r[5]
    print("%3d: %6d %6d %6d %6d %6d | %6d %6d %6d | %6d" % (I, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[11] - r[8], r[9] - r[11], r[10] - r[9], nxt - r[0]))
print("total cycles of the rows:", st[nb - 1][5] - st[0][0])
