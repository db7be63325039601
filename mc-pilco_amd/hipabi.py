"""ctypes binding of libmcpilco_hip.so -- the C ABI declared in include/mcpilco_hip.h.

The library is the product's only compute path: if it is missing or does not load, importing
this module's ``lib()`` raises (there is no CPU fallback anywhere in the package).
PyTorch is used for device memory and streams only: tensors are passed as ``data_ptr()``
and launches go to ``torch.cuda.current_stream()``.
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmcpilco_hip.so")
# Kernel experiments (tools/, variant builds of mc-pilco_amd/build.py --variant*) may load another build: only when they ask for it by
# MCPILCO_HIP_EXPERIMENT=1 next to MCPILCO_HIP_LIB -- a stray MCPILCO_HIP_LIB alone never swaps the product library.
if os.environ.get("MCPILCO_HIP_EXPERIMENT") == "1" and os.environ.get("MCPILCO_HIP_LIB"):
    LIB_PATH = os.environ["MCPILCO_HIP_LIB"]

MAX_GP, MAX_STATE, MAX_INPUT, MAX_GPDIM, MAX_PFEAT, MAX_BASIS, MAX_TRAIN = 8, 16, 8, 32, 32, 1024, 4096
OK = 0
ERRORS = {-1: "MCP_ERR_ARG", -2: "MCP_ERR_LIMIT", -3: "MCP_ERR_WORKSPACE", -4: "MCP_ERR_LAUNCH", -5: "MCP_ERR_COMM"}
ABI_VERSION = 6
COMM_ID_BYTES = 128
STATUS_NAN, STATUS_NONPOS_VAR, STATUS_NOT_SPD, STATUS_SYNC = 1, 2, 4, 8
FWD_NO_GP_SHARDING = 2  # flag in mcp_rollout_fwd's particle_pred argument (MCP_FWD_NO_GP_SHARDING)
FWD_KT_PACKED, FWD_XJ_PACKED = 4, 8  # the workspace still holds the packed operand copies of an earlier call on the same model (MCP_FWD_*_PACKED)
POLICY_PLAIN, POLICY_ANGLES, POLICY_TRAJ = 0, 1, 2
COST_CARTPOLE, COST_TRAJ = 0, 1

dptr = C.c_void_p


class Kernel(C.Structure):
    _fields_ = [("D", C.c_int32), ("poly_deg", C.c_int32), ("lam", C.c_double), ("sigma_n2", C.c_double), ("mean", C.c_double),
                ("inv_ls", dptr), ("w1", dptr), ("w20", dptr), ("w21", dptr), ("scal", dptr)]


class GP(C.Structure):
    _fields_ = [("kern", Kernel), ("N", C.c_int32), ("Npad", C.c_int32), ("Xt", dptr), ("X", dptr), ("alpha", dptr), ("Kinv", dptr),
                ("aX", dptr)]


class Model(C.Structure):
    _fields_ = [("S", C.c_int32), ("U", C.c_int32), ("G", C.c_int32), ("D", C.c_int32), ("n_angle", C.c_int32),
                ("n_not_angle", C.c_int32), ("angle", C.c_int32 * MAX_STATE), ("not_angle", C.c_int32 * MAX_STATE),
                ("vel", C.c_int32 * MAX_GP), ("not_vel", C.c_int32 * MAX_GP), ("Ts", C.c_double), ("var_scale", C.c_double * MAX_GP),
                ("gp", GP * MAX_GP)]


class Meas(C.Structure):
    _fields_ = [("n", C.c_int32), ("pos", C.c_int32 * MAX_STATE), ("vel", C.c_int32 * MAX_STATE), ("std_pos", C.c_double * MAX_STATE),
                ("b0", C.c_double), ("b1", C.c_double), ("a0", C.c_double), ("a1", C.c_double), ("pos_noise", dptr), ("meas", dptr)]


class Policy(C.Structure):
    _fields_ = [("kind", C.c_int32), ("S", C.c_int32), ("P", C.c_int32), ("B", C.c_int32), ("U", C.c_int32), ("squash", C.c_int32),
                ("n_angle", C.c_int32), ("n_non_angle", C.c_int32), ("angle", C.c_int32 * MAX_STATE),
                ("non_angle", C.c_int32 * MAX_STATE), ("traj_len", C.c_int32), ("p_drop", C.c_double), ("log_ls", dptr),
                ("centers", dptr), ("weight", dptr), ("u_max", dptr), ("target_traj", dptr), ("bias", dptr), ("g_bias", dptr), ("meas", Meas)]


class Noise(C.Structure):
    _fields_ = [("eps", dptr), ("masks", dptr), ("seed", C.c_uint64), ("call", C.c_uint64), ("particle_offset", C.c_int64), ("call_dev", dptr)]


class OptState(C.Structure):
    _fields_ = [("step", C.c_int64), ("attempt", C.c_int64), ("pending", C.c_int64), ("adam_t", C.c_int64), ("total_attempts", C.c_int64),
                ("es2", C.c_double), ("cost_prev", C.c_double)]


OPT_MAX_ATTEMPTS, OPT_MAX_TENSORS, OPT_RECORD_DOUBLES = 10, 32, 12


class NllGP(C.Structure):
    _fields_ = [("log_ls", dptr), ("log_lambda", dptr), ("sigma_n_log", dptr), ("mean", dptr), ("mpk1", dptr), ("mpk2", dptr), ("Y", dptr),
                ("y_scale", C.c_double), ("sigma_n_num2", C.c_double), ("g_log_ls", dptr), ("g_log_lambda", dptr), ("g_sigma_n_log", dptr),
                ("g_mean", dptr), ("g_mpk1", dptr), ("g_mpk2", dptr), ("loss", dptr)]


class Cost(C.Structure):
    _fields_ = [("kind", C.c_int32), ("S", C.c_int32), ("angle_index", C.c_int32), ("pos_index", C.c_int32),
                ("target_angle", C.c_double), ("target_pos", C.c_double), ("ls_angle", C.c_double), ("ls_pos", C.c_double),
                ("n_used", C.c_int32), ("used", C.c_int32 * MAX_STATE), ("target_traj", dptr), ("lengthscales", dptr)]


class Dispatch(C.Structure):
    """include/mcpilco_hip_debug.h: struct mcp_dispatch -- the request a call carries (all zero = automatic) and what it reports back."""
    _fields_ = [("fwd_particles", C.c_int32), ("gp_sharding", C.c_int32), ("fwd_lean", C.c_int32), ("policy_split", C.c_int32), ("row_split", C.c_int32),
                ("cluster_map", C.c_int32), ("fwd_no_xlds", C.c_int32),
                ("fwd_gb", C.c_int32), ("bwd_particles", C.c_int32), ("bwd_lean", C.c_int32), ("bwd_pipe", C.c_int32), ("chol_form", C.c_int32), ("stamp_block", C.c_uint32),
                ("fwd_stamps", dptr), ("bwd_stamps", dptr), ("ran_particles", C.c_int32), ("ran_gp_sharded", C.c_int32), ("ran_fwd_lean", C.c_int32),
                ("ran_bwd_lean", C.c_int32), ("ran_row_split", C.c_int32), ("ran_bwd_pipe", C.c_int32)]


# The dispatch request of THIS PROCESS's calls through `ops` (all zero: automatic).  It lives here, in the host layer -- the library keeps no
# dispatch state; tests and tools set its fields through the `mcp_debug_*` methods of `lib()` (the names the library itself exported until
# round 4), every `ops` call passes it along and finds in it what ran.
DISPATCH = Dispatch()

_SIGS = {
    "mcp_abi_version": (C.c_int, []),
    "mcp_build_info": (C.c_char_p, []),
    "mcp_cov_build": (C.c_int, [C.POINTER(Kernel), C.c_int, dptr, C.c_int, dptr, C.c_int, dptr, C.c_int, dptr]),
    "mcp_cov_diag": (C.c_int, [C.POINTER(Kernel), C.c_int, dptr, C.c_int, dptr, dptr]),
    "mcp_chol_factor": (C.c_int, [C.c_int, dptr, C.c_int, dptr, dptr, dptr]),
    "mcp_chol_inverse": (C.c_int, [C.c_int, dptr, C.c_int, dptr, C.c_int, dptr, C.c_int, dptr]),
    "mcp_chol_factor_ex": (C.c_int, [C.c_int, dptr, C.c_int, dptr, dptr, dptr, C.POINTER(Dispatch)]),
    "mcp_chol_inverse_ex": (C.c_int, [C.c_int, dptr, C.c_int, dptr, C.c_int, dptr, C.c_int, dptr, C.POINTER(Dispatch)]),
    "mcp_sym_sandwich": (C.c_int, [C.c_int, dptr, C.c_int, dptr, C.c_int, dptr, C.c_int, dptr, dptr]),
    "mcp_gp_alpha": (C.c_int, [C.c_int, dptr, C.c_int, dptr, C.c_double, dptr, dptr]),
    "mcp_sod_workspace_bytes": (C.c_size_t, [C.c_int]),
    "mcp_sod_select": (C.c_int, [C.POINTER(Kernel), C.c_int, dptr, C.c_double, dptr, dptr, dptr, C.c_size_t, dptr]),
    "mcp_nll_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "mcp_nll_grad": (C.c_int, [C.POINTER(Kernel), C.c_int, dptr, dptr, C.c_int, dptr, dptr, dptr, C.c_size_t, dptr]),
    "mcp_gp_pack": (C.c_int, [C.c_int, C.c_int, dptr, dptr, dptr, C.c_int, C.c_int, dptr, dptr, dptr, dptr, dptr, dptr]),
    "mcp_posterior_fwd": (C.c_int, [C.POINTER(GP), C.c_int, dptr, dptr, dptr, dptr, dptr, dptr, dptr]),
    "mcp_posterior_fwd_ex": (C.c_int, [C.POINTER(GP), C.c_int, dptr, dptr, dptr, dptr, dptr, dptr, dptr, C.POINTER(Dispatch)]),
    "mcp_posterior_bwd": (C.c_int, [C.c_int, C.c_int, dptr, dptr, dptr, dptr, dptr, dptr]),
    "mcp_rollout_workspace_bytes": (C.c_size_t, [C.POINTER(Model), C.POINTER(Policy), C.c_int, C.c_int]),
    "mcp_rollout_fwd": (C.c_int, [C.POINTER(Model), C.POINTER(Policy), C.POINTER(Noise), C.c_int, C.c_int, C.c_int, dptr, dptr, dptr,
                                  dptr, dptr, dptr, C.c_size_t, dptr]),
    "mcp_rollout_bwd": (C.c_int, [C.POINTER(Model), C.POINTER(Policy), C.POINTER(Noise), C.c_int, C.c_int, dptr, dptr, dptr, dptr, dptr,
                                  dptr, dptr, dptr, dptr, dptr, C.c_size_t, dptr]),
    "mcp_rollout_fwd_ex": (C.c_int, [C.POINTER(Model), C.POINTER(Policy), C.POINTER(Noise), C.c_int, C.c_int, C.c_int, dptr, dptr, dptr,
                                     dptr, dptr, dptr, C.c_size_t, dptr, C.POINTER(Dispatch)]),
    "mcp_rollout_bwd_ex": (C.c_int, [C.POINTER(Model), C.POINTER(Policy), C.POINTER(Noise), C.c_int, C.c_int, dptr, dptr, dptr, dptr, dptr,
                                     dptr, dptr, dptr, dptr, dptr, C.c_size_t, dptr, C.POINTER(Dispatch)]),
    "mcp_cost_fwd": (C.c_int, [C.POINTER(Cost), C.c_int, C.c_int, dptr, dptr, dptr, dptr, dptr]),
    "mcp_cost_finalize": (C.c_int, [C.c_int, C.c_int, dptr, C.POINTER(C.c_int64), dptr, dptr]),
    "mcp_cost_bwd": (C.c_int, [C.POINTER(Cost), C.c_int, C.c_int, dptr, dptr, C.c_double, dptr, dptr]),
    "mcp_cost_sums": (C.c_int, [C.c_int, C.c_int, dptr, dptr, dptr, dptr]),
    "mcp_cost_finalize_sums": (C.c_int, [C.c_int, C.c_int64, dptr, dptr, dptr, dptr, dptr]),
    "mcp_adam_step_guarded": (C.c_int, [C.c_int, C.POINTER(dptr), C.POINTER(dptr), C.POINTER(dptr), C.POINTER(dptr), C.POINTER(C.c_int64), C.c_double,
                                        C.c_double, C.c_double, C.c_double, dptr, C.c_int64, C.c_int, dptr, dptr, dptr, dptr]),
    "mcp_nll_epoch_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "mcp_nll_epoch": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, dptr, dptr, dptr, C.c_size_t, dptr]),
    "mcp_policy_step_commit": (C.c_int, [dptr, C.c_int, dptr, dptr, dptr, dptr, dptr, dptr, dptr, dptr, C.c_double, C.c_double, C.c_double, C.c_int,
                                         dptr, dptr]),
    "mcp_comm_unique_id": (C.c_int, [C.c_char_p]),
    "mcp_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_char_p]),
    "mcp_comm_world": (C.c_int, []),
    "mcp_allreduce_grad": (C.c_int, [dptr, C.c_size_t, dptr]),
    "mcp_comm_destroy": (C.c_int, []),
}
EXPORTED = [k for k in _SIGS if not k.endswith("_ex")]        # include/mcpilco_hip.h
EXPORTED_DEBUG = [k for k in _SIGS if k.endswith("_ex")]      # include/mcpilco_hip_debug.h

_lib = None


def lib():
    """The loaded library; raises RuntimeError when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libmcpilco_hip.so is missing (%s): build it with `python mc-pilco_amd/build.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.mcp_abi_version() != ABI_VERSION:
            raise RuntimeError("libmcpilco_hip.so ABI version mismatch")
        _lib = _Lib(handle)
    return _lib


class _Lib:
    """The loaded library plus the `mcp_debug_*` names as HOST-SIDE accessors of DISPATCH (what the library exported as process-wide setters
    until round 4): tests and tools keep their calls, the state they set is this module's, and it reaches the kernels with each call."""

    def __init__(self, handle):
        self._h = handle

    def __getattr__(self, name):
        return getattr(self._h, name)

    # forward rollout: particles per workgroup 1 / 2 / 4 / 16, 0 = automatic
    def mcp_debug_set_particles_per_wg(self, p):
        DISPATCH.fwd_particles = int(p)

    def mcp_debug_last_particles_per_wg(self):
        return int(DISPATCH.ran_particles)

    def mcp_debug_set_bwd_particles(self, pb):
        DISPATCH.bwd_particles = int(pb)

    def mcp_debug_set_stamp_buffer(self, p):
        DISPATCH.fwd_stamps = p

    def mcp_debug_set_stamp_block(self, b):
        DISPATCH.stamp_block = max(0, int(b))

    def mcp_debug_set_fwd_mode(self, xlds, gb):  # xlds 0: never stage the small operands in LDS; gb: GPs per pass (0 = as many as fit)
        DISPATCH.fwd_no_xlds = 1 if int(xlds) == 0 else 0
        DISPATCH.fwd_gb = int(gb)

    def mcp_debug_set_bwd_stamp_buffer(self, p):
        DISPATCH.bwd_stamps = p

    def mcp_debug_last_bwd_pipe(self):
        return int(DISPATCH.ran_bwd_pipe)

    def mcp_debug_set_bwd_pipe(self, mode):  # -1 / 1 wherever it applies, 0 never
        DISPATCH.bwd_pipe = 1 if int(mode) == 0 else 0

    def mcp_debug_set_gp_sharding(self, mode):  # -1 automatic, 0 never, 1 whenever the grid fits the device
        DISPATCH.gp_sharding = {-1: 0, 0: 1, 1: 2}[int(mode)]

    def mcp_debug_set_policy_split(self, mode):  # -1 automatic, 0 off, 1 whenever the shape allows
        DISPATCH.policy_split = {-1: 0, 0: 1, 1: 2}[int(mode)]

    def mcp_debug_set_row_split(self, mode):  # -1 automatic, 0 off, 1 / 2 two row parts whenever the shape allows, 3 three
        DISPATCH.row_split = {-1: 0, 0: 1, 1: 2, 2: 2, 3: 3}[int(mode)]

    def mcp_debug_set_cluster_map(self, mode):  # -1 automatic, 0 the workgroups of a tile on one XCD, 1 row part major
        DISPATCH.cluster_map = {-1: 0, 0: 1, 1: 2}[int(mode)]

    def mcp_debug_last_row_split(self):
        return int(DISPATCH.ran_row_split)

    def mcp_debug_last_gp_sharded(self):
        return int(DISPATCH.ran_gp_sharded)

    def mcp_debug_set_fwd_lean(self, mode):  # -1 / 1 wherever it applies, 0 never
        DISPATCH.fwd_lean = 1 if int(mode) == 0 else 0

    def mcp_debug_last_fwd_lean(self):
        return int(DISPATCH.ran_fwd_lean)

    def mcp_debug_set_chol_mfma(self, form):  # 1 default, 0 the round-1/2 kernels, 2 the round-3 forms, 3 one-wave inverse columns
        DISPATCH.chol_form = {1: 0, 0: 1, 2: 2, 3: 3}[int(form)]

    def mcp_debug_set_bwd_lean(self, mode):  # -1 automatic, 0 never
        DISPATCH.bwd_lean = 1 if int(mode) == 0 else 0

    def mcp_debug_last_bwd_lean(self):
        return int(DISPATCH.ran_bwd_lean)


def check(rc, what):
    if rc != OK:
        raise RuntimeError("%s failed: %s" % (what, ERRORS.get(rc, rc)))


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libmcpilco_hip operates on GPU memory only (got a %s tensor); there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("non-contiguous tensor passed to the HIP ABI")
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """The current HIP stream of the current device (every launch goes there).  Through torch's raw-handle getter where this build has
    it: ``torch.cuda.current_stream()`` builds a Stream object per call, ~11 us -- 3-5 of them per optimizer step / training epoch."""
    if _raw_stream is not None and _cur_device is not None:
        return C.c_void_p(_raw_stream(_cur_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def f64(t, device):
    return torch.as_tensor(t, dtype=torch.float64).to(device).contiguous()
