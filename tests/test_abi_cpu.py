"""CPU-only checks of the drop-in boundary: the C-ABI library is built, loads, exports every
symbol include/mcpilco_hip.h declares, its structs have the layout the ctypes binding assumes,
and the host-side argument validation rejects bad descriptors without touching a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mcpilco_hip.h")


DEBUG_HEADER = os.path.join(ROOT, "include", "mcpilco_hip_debug.h")


def declared_symbols(header=HEADER):
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mcp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from mc_pilco_amd import hipabi

    lib = hipabi.lib()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "libmcpilco_hip.so does not export %s" % n
    assert set(hipabi.EXPORTED) == set(names)
    # the test / diagnostic entry points live in their own header, outside the boundary, and the binding knows exactly those: each is its plain
    # namesake plus a per-call mcp_dispatch (round 5: the library exports no setter and keeps no dispatch state)
    dbg = declared_symbols(DEBUG_HEADER)
    assert dbg and all(n.endswith("_ex") and n[:-3] in names for n in dbg)
    for n in dbg:
        assert hasattr(lib, n)
    assert set(hipabi.EXPORTED_DEBUG) == set(dbg)
    import ctypes

    raw = ctypes.CDLL(hipabi.LIB_PATH)
    for n in ("mcp_debug_set_particles_per_wg", "mcp_debug_set_gp_sharding", "mcp_debug_set_bwd_particles", "mcp_debug_set_chol_mfma",
              "mcp_debug_last_fwd_lean"):
        assert not hasattr(raw, n), "the library still exports the process-wide hook %s" % n
    # the header, the binding and the built library carry ONE version (a stale library or an old struct layout is rejected at load time)
    hdr = int(re.search(r"#define MCP_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert lib.mcp_abi_version() == hipabi.ABI_VERSION == hdr == 6
    assert b"gfx950" in lib.mcp_build_info()


def test_struct_layout_matches_header(tmp_path):
    from mc_pilco_amd import hipabi

    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "mcpilco_hip.h"\n'
        "int main(){printf(\"%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu \", sizeof(mcp_kernel), sizeof(mcp_gp), sizeof(mcp_model),"
        " sizeof(mcp_policy), sizeof(mcp_noise), sizeof(mcp_cost), offsetof(mcp_model, gp), offsetof(mcp_policy, log_ls),"
        " offsetof(mcp_cost, target_traj), sizeof(mcp_meas), offsetof(mcp_policy, meas), offsetof(mcp_meas, pos_noise)); "
        "printf(\"%zu %zu %zu %d %d %d\\n\", sizeof(mcp_opt_state), offsetof(mcp_opt_state, es2), offsetof(mcp_opt_state, total_attempts),"
        " MCP_OPT_MAX_ATTEMPTS, MCP_OPT_MAX_TENSORS, MCP_OPT_RECORD_DOUBLES); return 0;}\n"
    )
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(hipabi.Kernel), C.sizeof(hipabi.GP), C.sizeof(hipabi.Model), C.sizeof(hipabi.Policy), C.sizeof(hipabi.Noise),
            C.sizeof(hipabi.Cost), hipabi.Model.gp.offset, hipabi.Policy.log_ls.offset, hipabi.Cost.target_traj.offset,
            C.sizeof(hipabi.Meas), hipabi.Policy.meas.offset, hipabi.Meas.pos_noise.offset,
            C.sizeof(hipabi.OptState), hipabi.OptState.es2.offset, hipabi.OptState.total_attempts.offset, hipabi.OPT_MAX_ATTEMPTS,
            hipabi.OPT_MAX_TENSORS, hipabi.OPT_RECORD_DOUBLES]
    assert got == want


def test_dispatch_struct_layout_matches_the_debug_header_and_every_object_depends_on_it(tmp_path):
    """`mcp_dispatch` (include/mcpilco_hip_debug.h) travels with every `_ex` call: the ctypes mirror must lay out every field where the header does
    (offset by offset), and -- round 6: a new field once left the forward's object on the old layout, its report words landed in what the backward read
    as the stamp pointer -- the build must count the header among the dependencies of EVERY object."""
    import re

    from mc_pilco_amd import build, hipabi

    hdr = open(os.path.join(ROOT, "include", "mcpilco_hip_debug.h")).read()
    body = hdr[hdr.index("typedef struct mcp_dispatch {"):hdr.index("} mcp_dispatch;")]
    names = re.findall(r"^\s*(?:int32_t|uint32_t|void\*)\s+(\w+);", body, re.M)
    assert names == [f[0] for f in hipabi.Dispatch._fields_]
    src = tmp_path / "dz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mcpilco_hip_debug.h"\nint main(){printf("%zu", sizeof(mcp_dispatch));'
                   + "".join('printf(" %%zu", offsetof(mcp_dispatch, %s));' % n for n in names) + "return 0;}\n")
    exe = tmp_path / "dz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(hipabi.Dispatch)] + [getattr(hipabi.Dispatch, n).offset for n in names]
    assert any(h.endswith("mcpilco_hip_debug.h") for h in build.HEADERS) and any(h.endswith("mcpilco_hip.h") for h in build.HEADERS)


def test_argument_validation_without_gpu():
    """Bad descriptors are rejected on the host (negative MCP_ERR_*), before any launch."""
    from mc_pilco_amd import hipabi

    lib = hipabi.lib()
    assert lib.mcp_cov_build(None, 4, None, 4, None, 0, None, 4, None) == -1
    assert lib.mcp_chol_factor(0, None, 0, None, None, None) == -1
    m, p, n = hipabi.Model(), hipabi.Policy(), hipabi.Noise()
    assert lib.mcp_rollout_fwd(C.byref(m), C.byref(p), C.byref(n), 4, 3, 1, None, None, None, None, None, None, 0, None) == -1
    assert lib.mcp_rollout_workspace_bytes(C.byref(m), C.byref(p), 0, 0) == 0
    p.P, p.B, p.U = 5, 200, 1
    assert lib.mcp_rollout_workspace_bytes(C.byref(m), C.byref(p), 400, 150) == 8 * (5 + 200 * 5 + 200 + 1) * 400  # (+ U: dJ/dbias)
    assert lib.mcp_sod_workspace_bytes(200) == 8 * (200 * 200 + 2 * 200)  # (W + the candidates' running sums and prior variances)
    # from 256 candidates on: + the multi-workgroup kernel's exchange granules (G = 5 workgroups x 2 parities x (4 + 2 N)) and the Gram matrix
    assert lib.mcp_sod_workspace_bytes(300) == 8 * (300 * 300 + 2 * 300) + 8 * (5 * 2 * 4 + 5 * 2 * 300 * 2) + 8 * 300 * 300
    # wide models (more than 15 GP-input dimensions): the forward part also holds the packed phase-J operands of the 16-particle kernel,
    # per GP 2 variants x 2 row tiles x Npad/8 pair-steps x 64 lanes x 2 doubles behind the hand-off granules
    m.G, m.D = 6, 24
    for g in range(6):
        m.gp[g].Npad = 400
    p.P, p.B, p.U = 24, 8, 6  # (tiny policy: the backward slabs stay below the forward part)
    xch = ((16 + 16) * 2 * 6 * 2 * 8 + 15) & ~15
    rxch = 1 * 2 * 6 * 2 * 16 * 25 * 4 * 8  # (the row-split cluster's partial sums -- tiles x 2 x G x 2 senders x 16 x (D + 1) x 2 values x 2 granules)
    assert lib.mcp_rollout_workspace_bytes(C.byref(m), C.byref(p), 16, 5) == max(xch + 6 * 2 * 2 * 50 * 64 * 2 * 8 + rxch, 8 * (24 + 8 * 24 + 6 * 8 + 6) * 16)
    c = hipabi.Cost()
    c.kind, c.S = 7, 4
    assert lib.mcp_cost_fwd(C.byref(c), 3, 4, None, None, None, None, None) == -1


def test_comm_wrapper_without_a_communicator():
    """mcp_allreduce_grad is a thin RCCL wrapper with one communicator per process: before mcp_comm_init it must refuse
    (MCP_ERR_COMM), never crash; RCCL itself is bound at run time (no link-time dependency)."""
    from mc_pilco_amd import hipabi

    lib = hipabi.lib()
    assert lib.mcp_comm_world() == 0
    assert lib.mcp_allreduce_grad(C.c_void_p(8), 4, None) == -5
    assert lib.mcp_allreduce_grad(None, 4, None) == -1
    assert lib.mcp_comm_init(0, 0, None) == -1
    assert lib.mcp_comm_destroy() == 0
    out = subprocess.check_output(["ldd", hipabi.LIB_PATH]).decode()
    assert "rccl" not in out


def test_product_refuses_cpu_tensors():
    """No CPU fallback: handing the operators a CPU tensor raises instead of computing."""
    import torch

    from mc_pilco_amd import ops

    sp = ops.KernelSpec(torch.ones(3, dtype=torch.float64), 1.0, 0.01)
    with pytest.raises(RuntimeError):
        ops.cov_build(sp, torch.zeros(4, 3, dtype=torch.float64))


def test_missing_library_fails_loudly(tmp_path):
    code = (
        "import sys; sys.path.insert(0, %r); import mcp_boot\n"
        "from mc_pilco_amd import hipabi\n"
        "hipabi.LIB_PATH = %r\n"
        "try:\n    hipabi.lib()\nexcept RuntimeError as e:\n    print('RAISED', 'no CPU fallback' in str(e).lower() or 'fallback' in str(e))\n"
    ) % (ROOT, str(tmp_path / "nope.so"))
    out = subprocess.check_output([sys.executable, "-c", code]).decode()
    assert "RAISED True" in out
