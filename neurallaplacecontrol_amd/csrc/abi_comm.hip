// Host side of libnlc_hip.so, collective unit: the optional library-owned RCCL communicator over the ranks of a K-sharded
// planner (include/nlc.h, multi-GPU note).  RCCL is bound at run time: the library links against HIP only, and a process that
// already holds an RCCL (the one PyTorch-ROCm ships, same soname) must not get a second copy.
#include "nlc_host.h"

using namespace nlc;
using namespace nlc::host;

namespace nlc {
namespace host {
static void bind_rccl(Rccl& r) {
  void* h = nullptr;
  std::string last;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);  // already in the process (torch.distributed's)?
    if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
    const char* e = dlerror();  // ONE call: dlerror() clears the message it returns
    if (e) last = e;
  }
  if (!h) {
    r.why = std::string("librccl.so.1 not found: ") + last;
    return;
  }
  r.GetUniqueId = (int (*)(Rccl::UniqueId*))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (int (*)(void**, int, Rccl::UniqueId, int))dlsym(h, "ncclCommInitRank");
  r.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
  r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
  r.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
  if (!r.ok) r.why = "librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
}
Rccl* rccl() {
  // bound once, thread-safely (function-local static initialisation): ctxs of different threads may ask at the same time
  static Rccl* r = [] {
    static Rccl inst;
    bind_rccl(inst);
    return &inst;
  }();
  return r;
}
}  // namespace host
}  // namespace nlc

extern "C" int nlc_comm_unique_id(void* id_out) {
  if (!id_out) return NLC_ERR_BAD_ARG;
  Rccl* r = rccl();
  if (!r->ok) {
    g_create_error = r->why;
    return NLC_ERR_UNSUPPORTED;
  }
  Rccl::UniqueId id;
  const int rc = r->GetUniqueId(&id);
  if (rc != 0) {
    g_create_error = std::string("ncclGetUniqueId: ") + r->GetErrorString(rc);
    return NLC_ERR_COMM;
  }
  std::memcpy(id_out, id.internal, NLC_COMM_ID_BYTES);
  return NLC_OK;
}

extern "C" int nlc_comm_destroy(nlc_ctx* c) {
  if (!c) return NLC_ERR_BAD_ARG;
  hipSetDevice(c->device);
  if (c->comm) {
    hipStreamSynchronize(c->stream);
    rccl()->CommDestroy(c->comm);
    c->comm = nullptr;
  }
  if (c->comm_gather) hipFree(c->comm_gather);
  c->comm_gather = nullptr;
  c->comm_gather_n = 0;
  c->comm_world = 0;
  return NLC_OK;
}

extern "C" int nlc_comm_init(nlc_ctx* c, int rank, int world, const void* id) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!id || world < 1 || rank < 0 || rank >= world) return fail(c, NLC_ERR_BAD_ARG, "nlc_comm_init: bad rank / world / id");
  Rccl* r = rccl();
  if (!r->ok) return fail(c, NLC_ERR_UNSUPPORTED, r->why);
  nlc_comm_destroy(c);
  NLC_HIP(c, hipSetDevice(c->device));
  Rccl::UniqueId uid;
  std::memcpy(uid.internal, id, NLC_COMM_ID_BYTES);
  void* comm = nullptr;
  const int rc = r->CommInitRank(&comm, world, uid, rank);
  if (rc != 0) return fail(c, NLC_ERR_COMM, std::string("ncclCommInitRank: ") + r->GetErrorString(rc));
  c->comm = comm;
  c->comm_world = world;
  c->comm_rank = rank;
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_comm_self_test(nlc_ctx* c) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->comm) return fail(c, NLC_ERR_STATE, "nlc_comm_self_test: no communicator (nlc_comm_init)");
  NLC_HIP(c, hipSetDevice(c->device));
  const int G = c->comm_world;
  double* dev = nullptr;
  NLC_HIP(c, hipMalloc((void**)&dev, (size_t)(G + 1) * sizeof(double)));
  const double mine = (double)(c->comm_rank + 1);
  hipError_t e = hipMemcpyAsync(dev + G, &mine, sizeof(double), hipMemcpyHostToDevice, c->stream);
  int rc = 0;
  if (e == hipSuccess) rc = rccl()->AllGather(dev + G, dev, 1, kNcclFloat64, c->comm, c->stream);
  std::vector<double> got((size_t)G, 0.0);
  if (e == hipSuccess && rc == 0) e = hipMemcpyAsync(got.data(), dev, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess && rc == 0) e = hipStreamSynchronize(c->stream);
  hipFree(dev);
  if (rc != 0) return fail(c, NLC_ERR_COMM, std::string("ncclAllGather: ") + rccl()->GetErrorString(rc));
  if (e != hipSuccess) return fail(c, NLC_ERR_HIP, std::string("nlc_comm_self_test: ") + hipGetErrorString(e));
  for (int g = 0; g < G; ++g)
    if (got[(size_t)g] != (double)(g + 1))
      return fail(c, NLC_ERR_COMM, "nlc_comm_self_test: the all-gather returned the wrong rank order / values");
  return NLC_OK;
  NLC_GUARD_END(c)
}
