// mcp_allreduce_grad: the one collective of a particle-sharded policy-gradient step -- an in-place
// all-reduce(sum) over xGMI of the flat fp64 message [dJ/dlog_ls | dJ/dcenters | dJ/dW | sum_m c_t | sum_m c_t^2 | flags]
// -- as a thin wrapper over RCCL with ONE communicator per process, created once (mcp_comm_init) and reused by every step.
// The reference has no collective anywhere (single process, MC_PILCO.py:522-525: backward, optimizer.step); this is the
// exchange that particle sharding adds between those two lines.
//
// RCCL is bound at run time (dlopen of librccl.so, the copy the process already has loaded when the host is PyTorch-ROCm):
// the library has no link-time dependency on it and single-GPU users never touch it.  Message sizes are 10-100 KB, i.e.
// latency-bound: nothing here is tuned for link bandwidth.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/mcpilco_hip.h"

namespace {

typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId_ {
  char internal[MCP_COMM_ID_BYTES];
};
typedef int (*fn_get_unique_id)(ncclUniqueId_*);
typedef int (*fn_comm_init_rank)(ncclComm_t*, int, ncclUniqueId_, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*fn_comm_destroy)(ncclComm_t);

const int kNcclFloat64 = 8;  // ncclDataType_t (rccl.h)
const int kNcclSum = 0;      // ncclRedOp_t

struct Rccl {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  ncclComm_t comm = nullptr;
  int world = 0, rank = -1;
};
Rccl g_rccl;

bool rccl_load() {
  if (g_rccl.handle) return true;
  // first the copy the process has loaded already (PyTorch-ROCm ships its own librccl under torch/lib: binding a second RCCL
  // from /opt/rocm into the same process would give two sets of communicators), then a fresh load
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (h) break;
  }
  for (const char* n : names) {
    if (h) break;
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  }
  if (!h) return false;
  g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
  g_rccl.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
  g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
  if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce || !g_rccl.comm_destroy) {
    dlclose(h);
    return false;
  }
  g_rccl.handle = h;
  return true;
}

}  // namespace

extern "C" int mcp_comm_unique_id(void* id_out) {
  if (!id_out) return MCP_ERR_ARG;
  if (!rccl_load()) return MCP_ERR_COMM;
  ncclUniqueId_ id;
  if (g_rccl.get_unique_id(&id) != 0) return MCP_ERR_COMM;
  memcpy(id_out, id.internal, MCP_COMM_ID_BYTES);
  return MCP_OK;
}

extern "C" int mcp_comm_init(int world, int rank, const void* id) {
  if (!id || world <= 0 || rank < 0 || rank >= world) return MCP_ERR_ARG;
  if (g_rccl.comm) return (g_rccl.world == world && g_rccl.rank == rank) ? MCP_OK : MCP_ERR_ARG;  // created once
  if (!rccl_load()) return MCP_ERR_COMM;
  ncclUniqueId_ uid;
  memcpy(uid.internal, id, MCP_COMM_ID_BYTES);
  ncclComm_t c = nullptr;
  if (g_rccl.comm_init_rank(&c, world, uid, rank) != 0 || !c) return MCP_ERR_COMM;
  g_rccl.comm = c;
  g_rccl.world = world;
  g_rccl.rank = rank;
  return MCP_OK;
}

extern "C" int mcp_comm_world(void) { return g_rccl.comm ? g_rccl.world : 0; }

extern "C" int mcp_allreduce_grad(double* flat, size_t n, void* stream) {
  if (!flat || n == 0) return MCP_ERR_ARG;
  if (!g_rccl.comm) return MCP_ERR_COMM;
  if (g_rccl.all_reduce(flat, flat, n, kNcclFloat64, kNcclSum, g_rccl.comm, (hipStream_t)stream) != 0) return MCP_ERR_COMM;
  return MCP_OK;
}

extern "C" int mcp_comm_destroy(void) {
  if (!g_rccl.comm) return MCP_OK;
  const int rc = g_rccl.comm_destroy(g_rccl.comm);
  g_rccl.comm = nullptr;
  g_rccl.world = 0;
  g_rccl.rank = -1;
  return rc == 0 ? MCP_OK : MCP_ERR_COMM;
}
