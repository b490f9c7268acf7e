// The MapReduce "reduce" across GPUs inside the library: RCCL all-reduce of the packed per-shard buffers
// (statistics_reducer, local_MapReduce.py:250-277; SURVEY.md section 8(b)3 `allreduce(ctx, phase)`).
//
// RCCL is resolved with dlopen at the first gp_comm_* call, so the library loads (and everything single-GPU works) on a host without it.
// A copy of librccl that is already in the process (torch ships one) is preferred: two RCCL runtimes in one process would each open the
// xGMI/IPC channels.  GPARML_RCCL_LIB overrides the search.  One communicator per context = one rank per GPU, created from a
// caller-supplied ncclUniqueId (rank 0 calls gp_comm_unique_id and hands the 128 bytes to the other ranks by whatever channel it has:
// MPI, a file, torch.distributed's store ...).  The collectives are enqueued on the context's stream: no host synchronisation, ordered
// behind phase 1 / phase 2 and in front of the global step / gp_finish like every other launch.
#include "gp_common.h"
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace gp {
namespace {
// the five RCCL entry points used, with the C types of rccl.h spelled out (ncclResult_t, ncclDataType_t, ncclRedOp_t are int-sized enums;
// ncclComm_t is an opaque pointer; ncclUniqueId is a 128-byte struct passed BY VALUE to ncclCommInitRank)
struct UniqueId { char internal[GP_COMM_ID_BYTES]; };
typedef int (*GetUniqueId_t)(UniqueId*);
typedef int (*CommInitRank_t)(void**, int, UniqueId, int);
typedef int (*AllReduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*CommDestroy_t)(void*);
typedef const char* (*GetErrorString_t)(int);
constexpr int kNcclFloat64 = 8, kNcclSum = 0;   // rccl.h: ncclFloat64 = ncclDouble = 8, ncclSum = 0

struct Rccl {
  void* handle = nullptr;
  GetUniqueId_t GetUniqueId = nullptr;
  CommInitRank_t CommInitRank = nullptr;
  AllReduce_t AllReduce = nullptr;
  CommDestroy_t CommDestroy = nullptr;
  GetErrorString_t GetErrorString = nullptr;
  std::string error;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  std::vector<std::string> names;
  if (const char* e = getenv("GPARML_RCCL_LIB")) names.push_back(e);
  for (const char* n : {"librccl.so", "librccl.so.1"}) names.push_back(n);
  void* h = nullptr;
  for (const auto& n : names) { h = dlopen(n.c_str(), RTLD_NOW | RTLD_NOLOAD); if (h) break; }   // a copy that is already mapped first
  if (!h) for (const auto& n : names) { h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL); if (h) break; }
  if (!h) for (const char* n : {"/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
  if (!h) { const char* m = dlerror(); g_rccl.error = std::string("librccl.so not found (set GPARML_RCCL_LIB): ") + (m ? m : ""); return; }
  g_rccl.GetUniqueId = (GetUniqueId_t)dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (CommInitRank_t)dlsym(h, "ncclCommInitRank");
  g_rccl.AllReduce = (AllReduce_t)dlsym(h, "ncclAllReduce");
  g_rccl.CommDestroy = (CommDestroy_t)dlsym(h, "ncclCommDestroy");
  g_rccl.GetErrorString = (GetErrorString_t)dlsym(h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) { g_rccl.error = "librccl.so lacks an expected ncclXxx symbol"; return; }
  g_rccl.handle = h;
}
int need_rccl(gp_ctx* c) {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.handle) return fail(c, GP_ERR_RCCL, "RCCL is not available: %s", g_rccl.error.c_str());
  return GP_OK;
}
int rccl_fail(gp_ctx* c, const char* what, int rc) {
  return fail(c, GP_ERR_RCCL, "%s failed: %s (ncclResult %d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?", rc);
}
}  // namespace

void comm_free(gp_ctx* c) {
  if (c->comm && g_rccl.handle) (void)g_rccl.CommDestroy(c->comm);
  c->comm = nullptr; c->comm_ranks = 0; c->comm_rank = -1;
}
}  // namespace gp

using namespace gp;

extern "C" int gp_comm_unique_id(void* id_out) {
  if (!id_out) return fail(nullptr, GP_ERR_BAD_ARG, "gp_comm_unique_id: NULL output");
  GP_TRY_RC(need_rccl(nullptr));
  UniqueId id;
  const int rc = g_rccl.GetUniqueId(&id);
  if (rc != 0) return rccl_fail(nullptr, "ncclGetUniqueId", rc);
  std::memcpy(id_out, id.internal, GP_COMM_ID_BYTES);
  return GP_OK;
}

extern "C" int gp_comm_init(gp_ctx* c, const void* unique_id, int nranks, int rank) {
  if (!c) return GP_ERR_BAD_ARG;
  if (!unique_id || nranks <= 0 || rank < 0 || rank >= nranks) return fail(c, GP_ERR_BAD_ARG, "gp_comm_init: bad id / nranks %d / rank %d", nranks, rank);
  GP_TRY_RC(need_rccl(c));
  GP_HIP(c, hipSetDevice(c->device));
  comm_free(c);
  UniqueId id;
  std::memcpy(id.internal, unique_id, GP_COMM_ID_BYTES);
  void* comm = nullptr;
  const int rc = g_rccl.CommInitRank(&comm, nranks, id, rank);
  if (rc != 0) return rccl_fail(c, "ncclCommInitRank", rc);
  c->comm = comm; c->comm_ranks = nranks; c->comm_rank = rank;
  return GP_OK;
}

extern "C" int gp_comm_available(void) { return need_rccl(nullptr); }

extern "C" int gp_comm_info(gp_ctx* c, int* nranks, int* rank, int64_t* stats_bytes, int64_t* grads_bytes, double* probe_sum) {
  if (!c) return GP_ERR_BAD_ARG;
  if (nranks) *nranks = c->comm ? c->comm_ranks : 0;
  if (rank) *rank = c->comm ? c->comm_rank : -1;
  if (stats_bytes) *stats_bytes = 8 * ((int64_t)c->M * (c->M + 1) / 2 + (int64_t)c->M * c->D + SC_COUNT);
  if (grads_bytes) *grads_bytes = 8 * ((int64_t)c->M * c->Q + c->Q);
  if (probe_sum) {
    if (!c->comm) return fail(c, GP_ERR_STATE, "gp_comm_info(probe) before gp_comm_init");
    GP_HIP(c, hipSetDevice(c->device));
    // one double per rank through the communicator; the last of the device scalars of the global step is free between evaluations
    double one = 1.0, *slot = c->gs + GS_COUNT - 1;
    GP_HIP(c, hipMemcpyAsync(slot, &one, sizeof(double), hipMemcpyHostToDevice, c->stream));
    const int rc = g_rccl.AllReduce(slot, slot, 1, kNcclFloat64, kNcclSum, c->comm, c->stream);
    if (rc != 0) return rccl_fail(c, "ncclAllReduce(probe)", rc);
    GP_HIP(c, hipMemcpyAsync(probe_sum, slot, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    GP_HIP(c, hipStreamSynchronize(c->stream));
  }
  return GP_OK;
}

extern "C" int gp_comm_destroy(gp_ctx* c) {
  if (!c) return GP_ERR_BAD_ARG;
  (void)hipSetDevice(c->device);
  comm_free(c);
  return GP_OK;
}

extern "C" int gp_allreduce(gp_ctx* c, int which) {
  if (!c) return GP_ERR_BAD_ARG;
  if (which != 0 && which != 1) return fail(c, GP_ERR_BAD_ARG, "gp_allreduce: which must be 0 (statistics) or 1 (gradient sums)");
  if (!c->comm) return fail(c, GP_ERR_STATE, "gp_allreduce before gp_comm_init");
  if (which == 0 && c->state < 1) return fail(c, GP_ERR_STATE, "gp_allreduce(statistics) before gp_phase1");
  if (which == 1 && c->state < 3) return fail(c, GP_ERR_STATE, "gp_allreduce(gradient sums) before gp_phase2");
  GP_HIP(c, hipSetDevice(c->device));
  if (which == 0) {
    // the statistics travel packed: Psi2's upper triangle | C | scalars (gp_stats_pack / gp_stats_unpack)
    GP_TRY_RC(gp_stats_pack(c));
    void* p = nullptr; int64_t n = 0;
    GP_TRY_RC(gp_stats_packed_buffer(c, &p, &n));
    const int rc = g_rccl.AllReduce(p, p, (size_t)n, kNcclFloat64, kNcclSum, c->comm, c->stream);
    if (rc != 0) return rccl_fail(c, "ncclAllReduce(statistics)", rc);
    GP_TRY_RC(gp_stats_unpack(c));
  } else {
    const size_t n = (size_t)c->M * c->Q + c->Q;
    const int rc = g_rccl.AllReduce(c->grads, c->grads, n, kNcclFloat64, kNcclSum, c->comm, c->stream);
    if (rc != 0) return rccl_fail(c, "ncclAllReduce(gradient sums)", rc);
  }
  return GP_OK;
}
