// libacrmi.so: multi-GPU - RCCL (dlopen) communicator and the all-gather of the result slots.
#include "acrmi_ctx.h"

// ---- multi-GPU: RCCL all-gather of the result slots -----------------------------------
// libacrmi.so has no link-time dependency on RCCL: the library is resolved at the first call - the copy the process
// already holds (PyTorch ships its own librccl.so) or the ROCm one.
namespace {
struct RcclId { char internal[128]; };   // ncclUniqueId (rccl.h:43)
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool tried = false;
};
Rccl g_rccl;
constexpr int RCCL_FLOAT32 = 7;          // ncclFloat32 (rccl.h:466)

bool rccl_load() {
  if (g_rccl.tried) return g_rccl.lib != nullptr;
  g_rccl.tried = true;
  const char* env = getenv("ACRMI_RCCL_LIB");
  const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)      // already mapped into the process?
    if (n && (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  for (const char* n : names) {
    if (h) break;
    if (n) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  }
  if (!h) return false;
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(h, "ncclAllGather"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GetErrorString)
    return false;
  g_rccl.lib = h;
  return true;
}
const char* rccl_err(int r) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"; }
}  // namespace

void comm_destroy(acrmi_ctx* c) {
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  c->comm = nullptr;
  c->comm_ranks = 0;
}

extern "C" {

int acrmi_comm_unique_id(void* id128) {
  if (!id128) return fail(nullptr, ACRMI_EINVAL, "acrmi_comm_unique_id: id is NULL");
  if (!rccl_load()) return fail(nullptr, ACRMI_ESTATE, "RCCL (librccl.so) could not be loaded: %s", dlerror());
  RcclId id;
  const int r = g_rccl.GetUniqueId(&id);
  if (r) return fail(nullptr, ACRMI_EHIP, "ncclGetUniqueId: %s", rccl_err(r));
  memcpy(id128, &id, sizeof id);
  return ACRMI_OK;
}

int acrmi_comm_init(acrmi_ctx* c, int n_ranks, int rank, const void* id128) {
  if (!c || !id128 || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(c, ACRMI_EINVAL, "acrmi_comm_init: bad arguments");
  if (!rccl_load()) return fail(c, ACRMI_ESTATE, "RCCL (librccl.so) could not be loaded: %s", dlerror());
  ON_DEVICE(c);
  comm_destroy(c);
  RcclId id;
  memcpy(&id, id128, sizeof id);
  const int r = g_rccl.CommInitRank(&c->comm, n_ranks, id, rank);
  if (r) {
    c->comm = nullptr;
    return fail(c, ACRMI_EHIP, "ncclCommInitRank(%d of %d): %s", rank, n_ranks, rccl_err(r));
  }
  c->comm_ranks = n_ranks;
  return ACRMI_OK;
}

int acrmi_comm_destroy(acrmi_ctx* c) {
  if (!c) return fail(c, ACRMI_EINVAL, "acrmi_comm_destroy: ctx is NULL");
  ON_DEVICE(c);
  comm_destroy(c);
  return ACRMI_OK;
}

int acrmi_allgather(acrmi_ctx* c, void* nccl_comm, const float* send_dev, float* recv_dev, size_t n_floats, void* stream) {
  if (!c || !send_dev || !recv_dev || n_floats == 0) return fail(c, ACRMI_EINVAL, "acrmi_allgather: bad arguments");
  void* comm = nccl_comm ? nccl_comm : c->comm;
  if (!comm) return fail(c, ACRMI_ESTATE, "acrmi_allgather: no communicator (acrmi_comm_init, or pass an ncclComm_t)");
  if (!rccl_load()) return fail(c, ACRMI_ESTATE, "RCCL (librccl.so) could not be loaded");
  ON_DEVICE(c);
  const int r = g_rccl.AllGather(send_dev, recv_dev, n_floats, RCCL_FLOAT32, comm, (hipStream_t)stream);
  if (r) return fail(c, ACRMI_EHIP, "ncclAllGather: %s", rccl_err(r));
  return ACRMI_OK;
}

}  // extern "C"
