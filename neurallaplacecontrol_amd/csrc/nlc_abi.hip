// Host side of libnlc_hip.so: context, weight repacking into MFMA fragment order, launch sequencing.
// See include/nlc.h for the contract of every entry point and the reference interface it replaces.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "nlc_kernels.h"
#include "nlc_pack.h"

namespace nlc {
int nl_pick_nt3(int need);
}

using namespace nlc;

namespace {

thread_local std::string g_create_error;

struct ProfEntry {
  std::string name;
  double total_ms = 0.0;
  int64_t launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct DeviceArena {
  double* base = nullptr;
  size_t n = 0;
  std::vector<double> host;
  size_t push(const std::vector<double>& v) {
    // 64-double (512 B) alignment so every fragment row starts on a cache line
    const size_t off = (host.size() + 63) / 64 * 64;
    host.resize(off);
    host.insert(host.end(), v.begin(), v.end());
    return off;
  }
};

}  // namespace

struct nlc_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  hipDeviceProp_t prop;

  // model
  bool has_model = false;
  nlc_model_desc md{};
  int g = 0, S = 0, P = 0;
  DeviceArena arena;
  GruArgs gru{};   // weight pointers + normalisation filled in
  NlNetArgs net{}; // general-t variant (b1 = raw bias)
  std::vector<double> W1s_host, b1_host;  // for folding the constant sphere inputs at configure time
  int* slot_dev = nullptr;                // (8*nt3) layer-3 slot -> c*S + k (de Hoog path)
  double* lin_tab = nullptr;              // nodes / weights of the linear ILT algorithms, cached per (algo, terms)
  int lin_algo = -1, lin_S = 0;
  int* eidx_dev = nullptr;                // (d*S) inverse: term k of dim c -> slot (slot-major F of the planner path)

  // Delta-t RNN baseline model
  bool has_rnn = false;
  nlc_rnn_desc rd{};
  double* rnn_base = nullptr;  // packed weights (device)
  RnnArgs rnn{};
  RnnHead rnn_head{};

  // NODE baseline model
  bool has_node = false;
  nlc_node_desc nd{};
  double* node_base = nullptr;
  NodeNetArgs node{};
  int node_ht = 0;

  // planner
  bool has_mppi = false;
  nlc_mppi_desc pd{};
  double* U[2] = {nullptr, nullptr};
  int ucur = 0;
  double* b1fold = nullptr;   // (h) device
  double* b1fold_fwd = nullptr;          // (h) device: the same fold for nlc_model_forward_const_t's query time
  double fwd_tn = -1.0;                  // normalised time b1fold_fwd was folded for
  std::vector<double> fwd_fold_host;     // its host copy (kept alive for the asynchronous upload)
  double* small = nullptr;    // action (<= T*nu) + beta_eta (2)
  double tn = 0.0;
  int nblk = 0;

  bool profiling = false;
  std::vector<ProfEntry> prof;
  std::vector<hipEvent_t> event_pool;

  // pinned host staging for the per-command small transfers (state, action_buffer, action)
  double* pinned = nullptr;
  size_t pinned_n = 0;
  hipEvent_t stage_ev = nullptr;  // recorded after the staged H2D copies of a command
  // planner options (nlc_set_option)
  int opt_rollout_variant = 0;          // 0 auto, 1 wave-per-tile, 2 latency-split (two launches), 3 fused one-launch body
  int opt_fused_roll_cap = 0;           // 0 auto (one chain per 16-sample tile, at most one per CU)
  int opt_repfunc_split = 1;            // staged de Hoog planner: latency-split representation kernel (h = 128)
  int opt_gru_coop = -1;                // stand-alone GRU encodes: cooperative (one tile per workgroup) kernel 1 / 0, -1 auto
  int opt_fused_chain_first_tiles = -1; // tiles per wave a chain's workgroup encodes before it starts walking (-1 auto)
  int opt_fused_partner_tiles = -2;     // tiles per wave after which a chain's CU partner sleeps (-1: never, -2 auto)
  int64_t opt_fused_max_samples = 4096; // auto: populations up to this size take the fused body (one chain per CU at most)
  int fused_blocks_per_cu = -1;         // occupancy of the fused kernel's 4-per-CU instance (queried once)
  int fused_blocks_per_cu3 = -1;        // ... of its 3-per-CU instance
  int fused_occ_h = 0;                  // hidden width the two occupancies were queried for
  int opt_fused_blocks_per_cu = 0;      // 0 auto (3 while chains sit on at most half of the CUs, else 4), 3 or 4
  bool fused_lost = false;              // a fused command gave up (hand-off timeout): later commands take the two-launch body
  int64_t fused_fallbacks = 0;          // commands re-run on the two-launch body after such a timeout
  int opt_horizon_chunks = 1;           // Fourier planner, wave-per-tile body (K > 8192): GRU encode of later horizon chunks beside the rollout of earlier ones
  int opt_dehoog_gru_chunks = 0;        // staged de Hoog planner: GRU encode in this many horizon chunks beside the step chain (0 / 1: one launch up front)
  int opt_dehoog_gru_lds_pad = 49152;   // unused dynamic LDS of those chunk launches (bytes): 32 KB + 48 KB -> two workgroups per CU
  hipStream_t gru_stream = nullptr;
  std::vector<hipEvent_t> ev_gru;
  int opt_dehoog_streams = 0;           // staged de Hoog planner: parts of the population on streams of their own (0 auto)
  std::vector<hipStream_t> aux_streams;
  hipEvent_t ev_fork = nullptr;
  std::vector<hipEvent_t> ev_join;
  double opt_fused_tile_step_ratio = 0.0;  // > 0: the adaptive partner rule (measured slower: profiles/r3_fused_small_shard.md); 0 = static schedule
  int opt_host_spin = 1;                // nlc_mppi_finish with a host action pointer: spin on a pinned word the merge kernel
                                        // stores (1) instead of hipStreamSynchronize (0)
  unsigned long long host_seq = 0;      // sequence number of the last command handed to the spin protocol
  int opt_fused_inline = 3;             // fused body: sampling / bounding and the weight reduction inside the launch
  int64_t opt_fused_spin_limit = 1 << 18;  // polls (~2 us each) before a waiting wave of the fused body gives up (~0.5 s)
  int opt_fused_test_drop_tile = -1;    // tests only: this encoder tile is never published (forces the timeout path)
  int opt_linear_fused = 1;             // fixed Talbot / Stehfest models: LIN instances of the rollout kernels
                                        // (0: the staged path)
  double* cp_lin = nullptr;             // [2][2 nt3][64] device: w_re / t and -w_im / t coefficient fragments (configure time)
  std::vector<std::pair<int, int>> slot_elems;  // (dim, term) of every layer-3 slot (nlc_pack.h), kept from nlc_set_model
  int opt_fused_keep_sync = 0;          // tools only: the merge kernel leaves the sync block as the launch left it (timeline dumps)
  const void* sync_clean_ws = nullptr;  // workspace whose fused sync block the last merge kernel left zeroed
  bool sync_dirty = false;              // a fused launch has used the sync block since
  // the last command's inputs, kept for a re-run on the two-launch body (nlc_mppi_finish, after a fused timeout)
  struct LastCommand {
    bool valid = false, inline_inputs = false, fused = false;
    int state_per_sample = 0, rng = 0;
    uint64_t seed = 0, counter = 0;
    double state_in[NLC_MAX_D] = {0};
    double abuf_in[kMaxInlineAbuf] = {0};
  } last;
  // optional native collective (nlc_comm_init): an RCCL communicator over the ranks of a K-sharded planner
  void* comm = nullptr;
  int comm_world = 0, comm_rank = 0;
  double* comm_gather = nullptr;  // (world, E, 2+T*nu) receive buffer of the per-command all-gather
  size_t comm_gather_n = 0;
};

namespace {

int fail(nlc_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

// Stand-alone GRU encode: the cooperative kernel (one 16-window tile per workgroup, gru_encode_coop_kernel) has a fraction
// of the latency of the wave-per-tile one at every width.  Measured on the MI355X (tools/gru_coop_probe.py):
//   g = 64 (hidden_units 128): better up to ~40 k windows (0.030 vs 0.093 ms at 4096, 0.233 vs 0.271 ms at 40960), level
//           at 80 k, 2 % slower from 160 k on (3.21 vs 3.14 ms at 655360);
//   g = 32 (hidden_units 64, two of the four waves idle): better up to ~4 k windows (0.023 vs 0.037 ms), 30-45 % slower
//           beyond 40 k;
//   g = 128 (hidden_units 256; 64 KB of images: two workgroups per CU instead of one): better at every size (0.079 vs
//           0.289 ms at 16 windows, 10.9 vs 11.7 ms at 655360).
bool gru_use_coop(const nlc_ctx* c, int64_t n_windows) {
  if (c->opt_gru_coop >= 0) return c->opt_gru_coop != 0;
  if (c->g == 128) return true;
  return n_windows <= (c->g == 32 ? 8192 : 50000);
}

#define NLC_HIP(c, expr)                                                                          \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      return fail((c), NLC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
  } while (0)

#define NLC_GUARD_BEGIN try {
#define NLC_GUARD_END(c)                                                       \
  }                                                                            \
  catch (const std::exception& e) {                                            \
    return fail((c), NLC_ERR_STATE, std::string("exception: ") + e.what());    \
  }                                                                            \
  catch (...) {                                                                \
    return fail((c), NLC_ERR_STATE, "unknown exception");                      \
  }

// ---- profiling: hipEvent pair around one launch, on the launch stream
struct ProfScope {
  nlc_ctx* c;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ProfEntry* entry = nullptr;
  hipEvent_t take_event() {
    if (!c->event_pool.empty()) {
      hipEvent_t e = c->event_pool.back();
      c->event_pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
  }
  hipStream_t st;
  ProfScope(nlc_ctx* ctx, const char* name, hipStream_t stream = nullptr, bool use_given = false)
      : c(ctx), st(use_given ? stream : ctx->stream) {
    if (!c->profiling) return;
    for (auto& p : c->prof)
      if (p.name == name) entry = &p;
    if (!entry) {
      c->prof.push_back(ProfEntry{});
      c->prof.back().name = name;
      entry = &c->prof.back();
    }
    e0 = take_event();
    e1 = take_event();
    hipEventRecord(e0, st);
  }
  ~ProfScope() {
    if (!entry) return;
    hipEventRecord(e1, st);
    entry->pending.emplace_back(e0, e1);
    entry->launches += 1;
  }
};

void prof_flush(nlc_ctx* c) {
  for (auto& p : c->prof) {
    for (auto& ev : p.pending) {
      hipEventSynchronize(ev.second);
      float ms = 0.f;
      hipEventElapsedTime(&ms, ev.first, ev.second);
      p.total_ms += ms;
      c->event_pool.push_back(ev.first);
      c->event_pool.push_back(ev.second);
    }
    p.pending.clear();
  }
}

struct Blob {
  const double* p;
  int64_t left;
  const double* take(int64_t n) {
    if (n > left) throw std::runtime_error("weight blob too short");
    const double* r = p;
    p += n;
    left -= n;
    return r;
  }
};

int64_t blob_size(const nlc_model_desc* d) {
  const int64_t g = d->h / 2, S = d->ilt.terms, P = d->d + 2, h = d->h;
  return 3 * g * d->nin + 3 * g * g + 6 * g + 3 * g * g + 3 * g * g + 6 * g + 2 * g + 2 + h * (2 * S + P) + h +
         h * h + h + 2 * d->d * S * h + 2 * d->d * S;
}

}  // namespace
static void linear_tables_host(int algo, int S, std::vector<double>& h);
namespace {
bool is_device_ptr(const void* p) {
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // plain (unregistered) host memory: not an error for us
    return false;
  }
  return at.type == hipMemoryTypeDevice;
}

void sphere_inputs(const nlc_ilt_desc& ilt, double tn, std::vector<double>& sph) {
  // [theta_s(0..S-1) | phi_s(0..S-1)] of s_k = gamma + i pi k / T (Fourier, de Hoog) or s_k = node_k / t (linear algorithms)
  const int S = ilt.terms;
  sph.assign(2 * S, 0.0);
  if (ilt.algo == NLC_ILT_FIXED_TALBOT || ilt.algo == NLC_ILT_STEHFEST) {
    std::vector<double> tab;
    linear_tables_host(ilt.algo, S, tab);
    for (int k = 0; k < S; ++k) {
      const double re = tab[k] / tn, im = tab[S + k] / tn;
      sph[k] = std::atan2(im, re);
      const double a2 = re * re + im * im;
      sph[S + k] = std::asin((a2 - 1.0) / (a2 + 1.0));
    }
    return;
  }
  const double Tt = ilt.scale * tn;
  const double gamma = ilt.alpha - std::log(ilt.tol) / (ilt.scale * Tt);
  for (int k = 0; k < S; ++k) {
    const double im = M_PI * (double)k / Tt;
    sph[k] = std::atan2(im, gamma);
    const double a2 = gamma * gamma + im * im;
    sph[S + k] = std::asin((a2 - 1.0) / (a2 + 1.0));
  }
}

}  // namespace

// =================================================================================== context
extern "C" int nlc_abi_version(void) { return NLC_ABI_VERSION; }

extern "C" int nlc_create(int device, nlc_ctx** out) {
  if (!out) {
    g_create_error = "nlc_create: out is NULL";
    return NLC_ERR_BAD_ARG;
  }
  *out = nullptr;
  try {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
      g_create_error = std::string("no HIP device: ") + hipGetErrorString(e);
      return NLC_ERR_HIP;
    }
    if (device < 0 || device >= n) {
      g_create_error = "device index out of range";
      return NLC_ERR_BAD_ARG;
    }
    nlc_ctx* c = new nlc_ctx();
    c->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&c->prop, device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
      g_create_error = std::string("HIP init failed: ") + hipGetErrorString(e);
      delete c;
      return NLC_ERR_HIP;
    }
    if (std::string(c->prop.gcnArchName).rfind("gfx950", 0) != 0) {
      g_create_error = std::string("libnlc_hip is built for gfx950 only, device is ") + c->prop.gcnArchName;
      hipStreamDestroy(c->own_stream);
      delete c;
      return NLC_ERR_UNSUPPORTED;
    }
    c->stream = c->own_stream;
    *out = c;
    return NLC_OK;
  } catch (...) {
    g_create_error = "exception in nlc_create";
    return NLC_ERR_STATE;
  }
}

// ---- RCCL, bound at run time: the library links against HIP only, and a process that already holds an RCCL (the one
// PyTorch-ROCm ships, same soname) must not get a second copy.
namespace {
struct Rccl {
  typedef struct { char internal[NLC_COMM_ID_BYTES]; } UniqueId;  // = ncclUniqueId (rccl.h: 128 opaque bytes)
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;  // the id is passed BY VALUE (rccl.h)
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;
  bool ok = false;
};
Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return &r;
  tried = true;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);  // already in the process (torch.distributed's)?
    if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    r.why = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "");
    return &r;
  }
  r.GetUniqueId = (int (*)(Rccl::UniqueId*))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (int (*)(void**, int, Rccl::UniqueId, int))dlsym(h, "ncclCommInitRank");
  r.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
  r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
  r.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
  if (!r.ok) r.why = "librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
  return &r;
}
constexpr int kNcclFloat64 = 8;  // rccl.h: ncclFloat64 = ncclDouble = 8
}  // namespace

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

extern "C" void nlc_destroy(nlc_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  nlc_comm_destroy(c);
  prof_flush(c);
  if (c->arena.base) hipFree(c->arena.base);
  if (c->rnn_base) hipFree(c->rnn_base);
  if (c->node_base) hipFree(c->node_base);
  if (c->slot_dev) hipFree(c->slot_dev);
  if (c->eidx_dev) hipFree(c->eidx_dev);
  if (c->lin_tab) hipFree(c->lin_tab);
  for (int i = 0; i < 2; ++i)
    if (c->U[i]) hipFree(c->U[i]);
  if (c->b1fold) hipFree(c->b1fold);
  if (c->cp_lin) hipFree(c->cp_lin);
  if (c->b1fold_fwd) hipFree(c->b1fold_fwd);
  if (c->small) hipFree(c->small);
  if (c->pinned) hipHostFree(c->pinned);
  if (c->stage_ev) hipEventDestroy(c->stage_ev);
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  for (hipEvent_t e : c->ev_join) hipEventDestroy(e);
  for (hipStream_t s2 : c->aux_streams) hipStreamDestroy(s2);
  for (hipEvent_t e : c->ev_gru) hipEventDestroy(e);
  if (c->gru_stream) hipStreamDestroy(c->gru_stream);
  for (hipEvent_t e : c->event_pool) hipEventDestroy(e);
  hipStreamDestroy(c->own_stream);
  delete c;
}

extern "C" const char* nlc_last_error(const nlc_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

extern "C" int nlc_set_stream(nlc_ctx* c, void* s) {
  if (!c) return NLC_ERR_BAD_ARG;
  c->stream = (hipStream_t)s;
  return NLC_OK;
}

extern "C" int nlc_set_option(nlc_ctx* c, const char* name, double value) {
  if (!c) return NLC_ERR_BAD_ARG;
  if (!name) return fail(c, NLC_ERR_BAD_ARG, "NULL option name");
  const std::string n(name);
  if (n == "rollout_variant") {
    if (value < 0 || value > 3) return fail(c, NLC_ERR_BAD_ARG, "rollout_variant must be 0 (auto), 1, 2 or 3");
    c->opt_rollout_variant = (int)value;
    c->fused_lost = false;  // an explicit choice re-arms the fused body after a timeout
  } else if (n == "horizon_chunks") {
    if (value < 1 || value > 8 || value != (int)value) return fail(c, NLC_ERR_BAD_ARG, "horizon_chunks must be 1 .. 8");
    c->opt_horizon_chunks = (int)value;
  } else if (n == "dehoog_gru_chunks") {
    if (value < 0 || value > 8 || value != (int)value) return fail(c, NLC_ERR_BAD_ARG, "dehoog_gru_chunks must be 0 .. 8");
    c->opt_dehoog_gru_chunks = (int)value;
  } else if (n == "dehoog_gru_lds_pad") {
    if (value < 0 || value > 120000) return fail(c, NLC_ERR_BAD_ARG, "dehoog_gru_lds_pad must be in 0 .. 120000 bytes");
    c->opt_dehoog_gru_lds_pad = (int)value;
  } else if (n == "dehoog_streams") {
    if (value < 0 || value > 4 || value != (int)value) return fail(c, NLC_ERR_BAD_ARG, "dehoog_streams must be 0 (auto), 1, 2, 3 or 4");
    c->opt_dehoog_streams = (int)value;
  } else if (n == "fused_tile_step_ratio") {
    if (value < 0 || value > 64) return fail(c, NLC_ERR_BAD_ARG, "fused_tile_step_ratio must be in 0 .. 64 (0 = static schedule)");
    c->opt_fused_tile_step_ratio = value;
  } else if (n == "host_spin") {
    if (value != 0 && value != 1) return fail(c, NLC_ERR_BAD_ARG, "host_spin must be 0 or 1");
    c->opt_host_spin = (int)value;
  } else if (n == "fused_blocks_per_cu") {
    if (value != 0 && value != 3 && value != 4) return fail(c, NLC_ERR_BAD_ARG, "fused_blocks_per_cu must be 0 (auto), 3 or 4");
    c->opt_fused_blocks_per_cu = (int)value;
  } else if (n == "fused_inline") {
    if (value < 0 || value > 3) return fail(c, NLC_ERR_BAD_ARG, "fused_inline must be 0, 1 (= 3), or the bit mask 1 weights | 2 sampling");
    c->opt_fused_inline = (int)value == 1 ? 3 : (int)value;
  } else if (n == "fused_spin_limit") {
    if (value < 1 || value > 4.0e9) return fail(c, NLC_ERR_BAD_ARG, "fused_spin_limit must be in 1 .. 4e9");
    c->opt_fused_spin_limit = (int64_t)value;
  } else if (n == "linear_fused") {
    c->opt_linear_fused = value != 0.0;
  } else if (n == "fused_keep_sync") {
    c->opt_fused_keep_sync = value != 0.0;
  } else if (n == "fused_test_drop_tile") {
    if (value < -1) return fail(c, NLC_ERR_BAD_ARG, "fused_test_drop_tile must be >= -1");
    c->opt_fused_test_drop_tile = (int)value;
  } else if (n == "fused_roll_cap") {
    if (value < 0) return fail(c, NLC_ERR_BAD_ARG, "fused_roll_cap must be >= 0 (0 = auto)");
    c->opt_fused_roll_cap = (int)value;
  } else if (n == "fused_chain_first_tiles") {
    if (value < -1 || value > 64) return fail(c, NLC_ERR_BAD_ARG, "fused_chain_first_tiles must be in -1..64 (-1 = auto)");
    c->opt_fused_chain_first_tiles = (int)value;
  } else if (n == "fused_partner_tiles") {
    if (value < -2 || value > 64) return fail(c, NLC_ERR_BAD_ARG, "fused_partner_tiles must be in -2..64 (-2 = auto, -1 = never)");
    c->opt_fused_partner_tiles = (int)value;
  } else if (n == "repfunc_split") {
    if (value != 0 && value != 1) return fail(c, NLC_ERR_BAD_ARG, "repfunc_split must be 0 or 1");
    c->opt_repfunc_split = (int)value;
  } else if (n == "gru_coop") {
    if (value != 0 && value != 1 && value != -1) return fail(c, NLC_ERR_BAD_ARG, "gru_coop must be -1 (auto), 0 or 1");
    c->opt_gru_coop = (int)value;
  } else if (n == "fused_max_samples") {
    if (value < 0) return fail(c, NLC_ERR_BAD_ARG, "fused_max_samples must be >= 0");
    c->opt_fused_max_samples = (int64_t)value;
  } else {
    return fail(c, NLC_ERR_BAD_ARG, "unknown option: " + n);
  }
  return NLC_OK;
}

extern "C" int nlc_synchronize(nlc_ctx* c) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_HIP(c, hipStreamSynchronize(c->stream));
  return NLC_OK;
}

extern "C" int nlc_device_info(nlc_ctx* c, char* name, int name_len, int* cus, int* mhz, double* gib) {
  if (!c) return NLC_ERR_BAD_ARG;
  if (name && name_len > 0) {
    std::snprintf(name, name_len, "%s (%s)", c->prop.name, c->prop.gcnArchName);
  }
  if (cus) *cus = c->prop.multiProcessorCount;
  if (mhz) *mhz = c->prop.clockRate / 1000;
  if (gib) *gib = (double)c->prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0);
  return NLC_OK;
}

// =================================================================================== ILT
static int check_ilt(nlc_ctx* c, const nlc_ilt_desc* d) {
  if (!d) return fail(c, NLC_ERR_BAD_ARG, "ilt desc is NULL");
  if (d->terms < 1 || d->terms > kMaxTerms) return fail(c, NLC_ERR_BAD_SHAPE, "ilt terms out of range [1,129]");
  if (!(d->tol > 0.0) || !(d->scale > 0.0)) return fail(c, NLC_ERR_BAD_ARG, "ilt tol/scale must be positive");
  if (d->algo != NLC_ILT_FOURIER && d->algo != NLC_ILT_DEHOOG && d->algo != NLC_ILT_FIXED_TALBOT &&
      d->algo != NLC_ILT_STEHFEST)
    return fail(c, NLC_ERR_UNSUPPORTED, "ilt_algorithm: fourier, dehoog, fixed_tablot and stehfest are implemented");
  if (d->algo == NLC_ILT_STEHFEST && (d->terms % 2 != 0 || d->terms < 2 || d->terms > 20))
    return fail(c, NLC_ERR_UNSUPPORTED, "stehfest: ilt_reconstruction_terms must be even, 2 .. 20 (Salzer weights in float64)");
  if (d->algo == NLC_ILT_FIXED_TALBOT && d->terms < 2)
    return fail(c, NLC_ERR_UNSUPPORTED, "fixed_tablot: ilt_reconstruction_terms must be >= 2");
  return NLC_OK;
}

// nodes and weights of the linear algorithms (mpmath 1.3.0 calculus/inverselaplace.py: FixedTalbot.calc_laplace_parameter /
// calc_time_domain_solution, Stehfest._coeff), uploaded once per (algorithm, terms): [node_re | node_im | w_re | w_im]
static void linear_tables_host(int algo, int S, std::vector<double>& h) {
  h.assign((size_t)4 * S, 0.0);
  double *nr = h.data(), *ni = nr + S, *wr = ni + S, *wi = wr + S;
  if (algo == NLC_ILT_FIXED_TALBOT) {
    const int M = S;
    const double r = 2.0 * M / 5.0;
    nr[0] = r;
    wr[0] = 0.4 * std::exp(r) / 2.0;
    for (int k = 1; k < M; ++k) {
      const double th = k * M_PI / M, cot = 1.0 / std::tan(th);
      nr[k] = r * th * cot;
      ni[k] = r * th;
      const double e = 0.4 * std::exp(nr[k]), cr = std::cos(ni[k]), ci = std::sin(ni[k]);
      const double fi = th * (1.0 + cot * cot) - cot;  // factor 1 + i fi
      wr[k] = e * (cr - ci * fi);
      wi[k] = e * (ci + cr * fi);
    }
  } else {
    const int M = S, M2 = S / 2;
    auto fac = [](int n) {
      long double f = 1.0L;
      for (int i = 2; i <= n; ++i) f *= i;
      return f;
    };
    for (int k = 1; k <= M; ++k) {
      long double z = 0.0L;
      for (int j = (k + 1) / 2; j <= (k < M2 ? k : M2); ++j)
        z += std::pow((long double)j, M2) * fac(2 * j) / (fac(M2 - j) * fac(j) * fac(j - 1) * fac(k - j) * fac(2 * j - k));
      nr[k - 1] = k * M_LN2;
      wr[k - 1] = (double)(((k + M2) % 2 ? -1.0L : 1.0L) * z * (long double)M_LN2);
    }
  }
}
static int linear_tables(nlc_ctx* c, const nlc_ilt_desc* d, const double** tab) {
  const int S = d->terms;
  if (c->lin_tab && c->lin_algo == d->algo && c->lin_S == S) {
    *tab = c->lin_tab;
    return NLC_OK;
  }
  std::vector<double> h;
  linear_tables_host(d->algo, S, h);
  if (!c->lin_tab) NLC_HIP(c, hipMalloc((void**)&c->lin_tab, (size_t)4 * kMaxTerms * sizeof(double)));
  // a kernel of an earlier call may still read the old tables -- on ANY stream the ctx was bound to since (the Python
  // mirror rebinds it to torch's current stream every call): this rare path waits for the whole device
  NLC_HIP(c, hipDeviceSynchronize());
  NLC_HIP(c, hipMemcpy(c->lin_tab, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
  c->lin_algo = d->algo;
  c->lin_S = S;
  *tab = c->lin_tab;
  return NLC_OK;
}

extern "C" int nlc_ilt_rep_inputs(nlc_ctx* c, const nlc_ilt_desc* d, const double* p, const double* t, int t_batched,
                                  int64_t B, int64_t Tt, int P, double* out) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (int r = check_ilt(c, d)) return r;
  if (B < 0 || Tt < 0 || P < 0) return fail(c, NLC_ERR_BAD_SHAPE, "negative shape");
  if (B * Tt == 0) return NLC_OK;
  if (!p || !t || !out) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  RepInArgs a{p, t, out, B, Tt, P, d->terms, t_batched, d->alpha, std::log(d->tol), d->scale, nullptr, nullptr, 1.0};
  if (d->algo == NLC_ILT_FIXED_TALBOT || d->algo == NLC_ILT_STEHFEST) {
    const double* tab = nullptr;
    if (int r = linear_tables(c, d, &tab)) return r;
    a.node_re = tab;
    a.node_im = tab + d->terms;
  }
  ProfScope ps(c, "rep_inputs_kernel");
  NLC_HIP(c, launch_rep_inputs(a, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_ilt_reconstruct(nlc_ctx* c, const nlc_ilt_desc* d, const double* theta, const double* phi,
                                   const double* t, int64_t N, int dd, double* x) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (int r = check_ilt(c, d)) return r;
  if (N < 0 || dd < 1) return fail(c, NLC_ERR_BAD_SHAPE, "bad N or d");
  if (N == 0) return NLC_OK;
  if (!theta || !phi || !t || !x) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  IltArgs a{theta, phi, t, x, N, dd, d->terms, d->alpha, std::log(d->tol), d->scale, nullptr, nullptr, 1.0, 1, 0, 0, 0};
  if (d->algo == NLC_ILT_FIXED_TALBOT || d->algo == NLC_ILT_STEHFEST) {
    const double* tab = nullptr;
    if (int r = linear_tables(c, d, &tab)) return r;
    // the Fourier kernel's coalesced stream with the algorithm's per-term phase and weight (round 3); the one-thread-per-row
    // kernel remains for a term count the stream's tiling does not take
    a.lin_wr = tab + 2 * d->terms;
    a.lin_wi = tab + 3 * d->terms;
    hipError_t le;
    {
      ProfScope ps(c, "ilt_linear_stream_kernel");
      le = launch_ilt_fourier(a, c->stream);
    }
    if (le == hipErrorInvalidValue) {
      (void)hipGetLastError();
      IltLinArgs la{theta, phi, t, x, N, dd, d->terms, a.lin_wr, a.lin_wi};
      ProfScope ps(c, "ilt_linear_kernel");
      NLC_HIP(c, launch_ilt_linear(la, c->stream));
    } else {
      NLC_HIP(c, le);
    }
  } else if (d->algo == NLC_ILT_FOURIER) {
    ProfScope ps(c, "ilt_fourier_kernel");
    NLC_HIP(c, launch_ilt_fourier(a, c->stream));
  } else {
    if (d->terms < 3 || d->terms > 33 || d->terms % 2 == 0)
      return fail(c, NLC_ERR_UNSUPPORTED, "dehoog: ilt_reconstruction_terms must be odd, 3 .. 33 (2M+1 terms)");
    ProfScope ps(c, "ilt_dehoog_kernel");
    NLC_HIP(c, launch_ilt_dehoog(a, c->stream));
  }
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_ilt_reconstruct_backward(nlc_ctx* c, const nlc_ilt_desc* d, const double* theta, const double* phi,
                                            const double* t, const double* grad_x, int64_t N, int dd,
                                            double* grad_theta, double* grad_phi) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (int r = check_ilt(c, d)) return r;
  if (d->algo == NLC_ILT_DEHOOG && (d->terms < 3 || d->terms > 33 || d->terms % 2 == 0))
    return fail(c, NLC_ERR_UNSUPPORTED, "dehoog: ilt_reconstruction_terms must be odd, 3 .. 33 (2M+1 terms)");
  if (N < 0 || dd < 1) return fail(c, NLC_ERR_BAD_SHAPE, "bad N or d");
  if (N == 0) return NLC_OK;
  if (!theta || !phi || !t || !grad_x || !grad_theta || !grad_phi) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  if (d->algo == NLC_ILT_DEHOOG) {
    // the QD tape lives in stream-ordered scratch of this launch (no ctx state: calls on different streams do not share it)
    const int64_t bytes = ilt_dehoog_bwd_scratch_bytes(N, dd, d->terms, nullptr);
    void* scratch = nullptr;
    NLC_HIP(c, hipMallocAsync(&scratch, (size_t)bytes, c->stream));
    IltDehoogBwdArgs da{theta, phi, t, grad_x, grad_theta, grad_phi, N, dd, d->terms, d->alpha, std::log(d->tol), d->scale, 1.0, scratch};
    hipError_t le;
    {
      ProfScope ps(c, "ilt_dehoog_bwd_kernel");
      le = launch_ilt_dehoog_bwd(da, c->stream);
    }
    const hipError_t fe = hipFreeAsync(scratch, c->stream);
    NLC_HIP(c, le);
    NLC_HIP(c, fe);
    return NLC_OK;
  }
  if (d->algo == NLC_ILT_FIXED_TALBOT || d->algo == NLC_ILT_STEHFEST) {
    const double* tab = nullptr;
    if (int r = linear_tables(c, d, &tab)) return r;
    IltLinBwdArgs la{theta, phi, t, grad_x, grad_theta, grad_phi, N, dd, d->terms, tab + 2 * d->terms, tab + 3 * d->terms};
    ProfScope ps(c, "ilt_linear_bwd_kernel");
    NLC_HIP(c, launch_ilt_linear_bwd(la, c->stream));
    return NLC_OK;
  }
  IltBwdArgs a{theta, phi, t, grad_x, grad_theta, grad_phi, N, dd, d->terms, d->alpha, std::log(d->tol), d->scale, 0, 0};
  ProfScope ps(c, "ilt_fourier_bwd_kernel");
  NLC_HIP(c, launch_ilt_fourier_bwd(a, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}

// =================================================================================== model
extern "C" int64_t nlc_model_blob_size(const nlc_model_desc* d) { return d ? blob_size(d) : -1; }

extern "C" int nlc_set_model(nlc_ctx* c, const nlc_model_desc* d, const double* w, int64_t n) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!d || !w) return fail(c, NLC_ERR_BAD_ARG, "NULL desc or weights");
  if (int r = check_ilt(c, &d->ilt)) return r;
  // (fixed_tablot / stehfest models: staged all-HIP forward and planner, like de Hoog)
  if (d->h != 64 && d->h != 128 && d->h != 256)
    return fail(c, NLC_ERR_UNSUPPORTED, "hidden_units must be 64, 128 or 256 (the kernels are instantiated for these widths)");
  if (d->d < 1 || d->d > 6) return fail(c, NLC_ERR_UNSUPPORTED, "state_dim must be in 1..6");
  if (d->nin < 1 || d->nin > NLC_MAX_NIN) return fail(c, NLC_ERR_UNSUPPORTED, "GRU input dim must be in 1..3");
  const bool linear_algo = d->ilt.algo == NLC_ILT_FIXED_TALBOT || d->ilt.algo == NLC_ILT_STEHFEST;
  if (!linear_algo && d->ilt.scale != 2.0) return fail(c, NLC_ERR_UNSUPPORTED, "fused model path needs ILT scale == 2");
  if (n != blob_size(d)) return fail(c, NLC_ERR_BAD_SHAPE, "weight blob size mismatch");
  NLC_HIP(c, hipSetDevice(c->device));
  const int g = d->h / 2, S = d->ilt.terms, P = d->d + 2, h = d->h, dd = d->d, nin = d->nin;
  Blob b{w, n};
  const double* Wih0 = b.take(3 * g * nin);
  const double* Whh0 = b.take(3 * g * g);
  const double* bih0 = b.take(3 * g);
  const double* bhh0 = b.take(3 * g);
  const double* Wih1 = b.take(3 * g * g);
  const double* Whh1 = b.take(3 * g * g);
  const double* bih1 = b.take(3 * g);
  const double* bhh1 = b.take(3 * g);
  const double* Wo = b.take(2 * g);
  const double* bo = b.take(2);
  const double* W1 = b.take((int64_t)h * (2 * S + P));
  const double* b1 = b.take(h);
  const double* W2 = b.take((int64_t)h * h);
  const double* b2 = b.take(h);
  const double* W3 = b.take((int64_t)2 * dd * S * h);
  const double* b3 = b.take(2 * dd * S);

  DeviceArena ar;
  // ---- GRU: layer-0 input weights with the bias folded into input column 3 (x = [a_0..a_{nin-1}, 0.., 1])
  std::vector<double> Wih0b((size_t)3 * g * 4, 0.0);
  for (int r = 0; r < 3 * g; ++r) {
    for (int j = 0; j < nin; ++j) Wih0b[(size_t)r * 4 + j] = Wih0[(size_t)r * nin + j];
    Wih0b[(size_t)r * 4 + 3] = bih0[r] + (r < 2 * g ? bhh0[r] : 0.0);
  }
  const size_t o_Wih0 = ar.push(pack_gru_chunked(Wih0b.data(), 4, 4, g));
  const size_t o_Whh0 = ar.push(pack_gru_chunked(Whh0, g, g, g));
  const size_t o_Wih1 = ar.push(pack_gru_chunked(Wih1, g, g, g));
  const size_t o_Whh1 = ar.push(pack_gru_chunked(Whh1, g, g, g));
  const size_t o_Wo = ar.push(pack_A(Wo, g, g, identity_rows(2)));
  std::vector<double> bhn0(bhh0 + 2 * g, bhh0 + 3 * g), brz1(2 * g), bin1(bih1 + 2 * g, bih1 + 3 * g),
      bhn1(bhh1 + 2 * g, bhh1 + 3 * g);
  for (int r = 0; r < 2 * g; ++r) brz1[r] = bih1[r] + bhh1[r];
  const size_t o_bhn0 = ar.push(bhn0), o_brz1 = ar.push(brz1), o_bin1 = ar.push(bin1), o_bhn1 = ar.push(bhn1);

  // ---- representation MLP
  // layer 1 split: sphere-coordinate columns [0, 2S) and latent columns [2S, 2S+P)
  std::vector<double> W1s((size_t)h * 2 * S), W1p((size_t)h * 8, 0.0);
  for (int r = 0; r < h; ++r) {
    for (int j = 0; j < 2 * S; ++j) W1s[(size_t)r * 2 * S + j] = W1[(size_t)r * (2 * S + P) + j];
    for (int j = 0; j < P; ++j) W1p[(size_t)r * 8 + j] = W1[(size_t)r * (2 * S + P) + 2 * S + j];
  }
  const auto rowsh = identity_rows(h);
  const size_t o_W1s = ar.push(pack_A(W1s.data(), 2 * S, 2 * S, rowsh));
  const size_t o_W1p = ar.push(pack_A(W1p.data(), 8, 8, rowsh));
  const size_t o_b1 = ar.push(std::vector<double>(b1, b1 + h));
  const size_t o_W2 = ar.push(pack_A(W2, h, h, rowsh));
  const size_t o_b2 = ar.push(std::vector<double>(b2, b2 + h));
  // layer 3: slot layout + ILT coefficient matrix (nlc_pack.h)
  const int nt3 = nl_pick_nt3(ilt_tiles_needed(dd, S));
  if (nt3 < 0) return fail(c, NLC_ERR_UNSUPPORTED, "2*d*S too large for the fused kernel (max 25 output tiles)");
  const IltSlots slots = make_ilt_slots(dd, S, nt3);
  c->slot_elems = slots.elems;
  const int n_even_groups = slots.n_even_groups;
  std::vector<double> b3p((size_t)nt3 * 16, 0.0);
  for (size_t i = 0; i < slots.rowmap3.size(); ++i)
    if (slots.rowmap3[i] >= 0) b3p[i] = b3[slots.rowmap3[i]];
  const size_t o_W3 = ar.push(pack_A(W3, h, h, slots.rowmap3));
  const size_t o_b3 = ar.push(b3p);
  const size_t o_Cp = ar.push(slots.Cp);

  // ---- upload
  double* base = nullptr;
  NLC_HIP(c, hipMalloc((void**)&base, ar.host.size() * sizeof(double)));
  hipError_t e = hipMemcpy(base, ar.host.data(), ar.host.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    hipFree(base);
    return fail(c, NLC_ERR_HIP, std::string("weight upload: ") + hipGetErrorString(e));
  }
  {
    std::vector<int> slot((size_t)nt3 * 8, -1);
    for (size_t i = 0; i < slot.size(); ++i)
      if (slots.elems[i].first >= 0) slot[i] = slots.elems[i].first * S + slots.elems[i].second;
    if (c->slot_dev) hipFree(c->slot_dev);
    c->slot_dev = nullptr;
    NLC_HIP(c, hipMalloc((void**)&c->slot_dev, slot.size() * sizeof(int)));
    NLC_HIP(c, hipMemcpy(c->slot_dev, slot.data(), slot.size() * sizeof(int), hipMemcpyHostToDevice));
    std::vector<int> eidx((size_t)dd * S, 0);
    for (size_t i = 0; i < slot.size(); ++i)
      if (slot[i] >= 0) eidx[slot[i]] = (int)i;
    if (c->eidx_dev) hipFree(c->eidx_dev);
    c->eidx_dev = nullptr;
    NLC_HIP(c, hipMalloc((void**)&c->eidx_dev, eidx.size() * sizeof(int)));
    NLC_HIP(c, hipMemcpy(c->eidx_dev, eidx.data(), eidx.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  NLC_HIP(c, hipStreamSynchronize(c->stream));
  if (c->arena.base) hipFree(c->arena.base);
  c->arena = std::move(ar);
  c->arena.base = base;
  c->md = *d;
  c->g = g;
  c->S = S;
  c->P = P;
  c->W1s_host = std::move(W1s);
  c->b1_host.assign(b1, b1 + h);

  GruArgs& G = c->gru;
  G = GruArgs{};
  G.nin = nin;
  for (int j = 0; j < nin; ++j) {
    G.mean[j] = d->action_mean[j];
    G.std[j] = d->action_std[j];
  }
  G.Wih0p = base + o_Wih0;
  G.Whh0p = base + o_Whh0;
  G.Wih1p = base + o_Wih1;
  G.Whh1p = base + o_Whh1;
  G.Wop = base + o_Wo;
  G.bhn0 = base + o_bhn0;
  G.brz1 = base + o_brz1;
  G.bin1 = base + o_bin1;
  G.bhn1 = base + o_bhn1;
  G.bo[0] = bo[0];
  G.bo[1] = bo[1];

  NlNetArgs& N = c->net;
  N = NlNetArgs{};
  N.d = dd;
  N.S = S;
  N.h = h;
  N.nt3 = nt3;
  N.n_even_groups = n_even_groups;
  N.W1p = base + o_W1p;
  N.W1s = base + o_W1s;
  N.b1 = base + o_b1;
  N.W2p = base + o_W2;
  N.b2 = base + o_b2;
  N.W3p = base + o_W3;
  N.b3p = base + o_b3;
  N.Cp = base + o_Cp;
  for (int i = 0; i < NLC_MAX_D; ++i) {
    N.state_mean[i] = i < dd ? d->state_mean[i] : 0.0;
    N.state_std[i] = i < dd ? d->state_std[i] : 1.0;
  }
  N.alpha = d->ilt.alpha;
  N.log_tol = std::log(d->ilt.tol);
  N.scale = d->ilt.scale;
  N.time_div = d->time_div;
  c->has_model = true;
  c->fwd_tn = -1.0;  // the constant-time forward's folded bias belongs to the previous weights
  c->has_mppi = false;  // a planner configured against the previous weights must be re-configured
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_gru_encode(nlc_ctx* c, const double* window, int64_t N, int B, double* out) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->has_model) return fail(c, NLC_ERR_STATE, "nlc_set_model has not been called");
  if (N < 0 || B < 1) return fail(c, NLC_ERR_BAD_SHAPE, "bad N or B");
  if (N == 0) return NLC_OK;
  if (!window || !out) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  GruArgs a = c->gru;
  a.mode = 0;
  a.window = window;
  a.N = N;
  a.B = B;
  a.out = out;
  ProfScope ps(c, "gru_encode_kernel");
  NLC_HIP(c, launch_gru_encode(a, c->g, c->stream, gru_use_coop(c, a.N)));
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int64_t nlc_model_workspace_bytes(nlc_ctx* c, int64_t N) {
  if (N < 0) return -1;
  int64_t n = N * 2 + 64;  // GRU latents
  if (c && c->has_model && c->md.ilt.algo == NLC_ILT_DEHOOG) n += 2 * N * c->md.d * c->S + 64;  // F_k re/im
  if (c && c->has_model && (c->md.ilt.algo == NLC_ILT_FIXED_TALBOT || c->md.ilt.algo == NLC_ILT_STEHFEST))
    n += 2 * N * c->md.d * c->S + 2 * N * c->S + 128;  // theta / phi rows + per-row sphere inputs
  return n * (int64_t)sizeof(double);
}

extern "C" int nlc_model_forward_const_t(nlc_ctx* c, const double* obs, const double* window, double ts_pred, int64_t N,
                                         int B, double* out, void* ws) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->has_model) return fail(c, NLC_ERR_STATE, "nlc_set_model has not been called");
  if (c->md.ilt.algo != NLC_ILT_FOURIER) return fail(c, NLC_ERR_UNSUPPORTED, "constant-time forward: fourier models only");
  if (N < 0 || B < 1) return fail(c, NLC_ERR_BAD_SHAPE, "bad N or B");
  if (!(ts_pred > 0.0)) return fail(c, NLC_ERR_BAD_ARG, "ts_pred must be > 0");
  if (N == 0) return NLC_OK;
  if (!obs || !window || !out || !ws) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  const double tn = ts_pred / c->md.time_div;
  const int h = c->md.h, S = c->S;
  if (!c->b1fold_fwd) NLC_HIP(c, hipMalloc((void**)&c->b1fold_fwd, h * sizeof(double)));
  if (tn != c->fwd_tn) {
    // fold the constant sphere inputs of layer 1 into its bias, as nlc_mppi_configure does for the planner
    // an earlier upload may still read the host copy, and a forward launched on a PREVIOUSLY bound stream may still read
    // the folded bias: a new query time is rare, so wait for the whole device
    NLC_HIP(c, hipDeviceSynchronize());
    std::vector<double> sph;
    sphere_inputs(c->md.ilt, tn, sph);
    c->fwd_fold_host.resize(h);
    for (int r = 0; r < h; ++r) {
      double acc = c->b1_host[r];
      for (int j = 0; j < 2 * S; ++j) acc += c->W1s_host[(size_t)r * 2 * S + j] * sph[j];
      c->fwd_fold_host[r] = acc;
    }
    NLC_HIP(c, hipMemcpyAsync(c->b1fold_fwd, c->fwd_fold_host.data(), h * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->fwd_tn = tn;
  }
  double* pa = (double*)ws;
  {
    GruArgs a = c->gru;
    a.mode = 0;
    a.window = window;
    a.N = N;
    a.B = B;
    a.out = pa;
    ProfScope ps(c, "gru_encode_kernel");
    NLC_HIP(c, launch_gru_encode(a, c->g, c->stream, gru_use_coop(c, a.N)));
  }
  ForwardArgs f{};
  f.net = c->net;
  f.net.b1 = c->b1fold_fwd;
  f.N = N;
  f.obs = obs;
  f.pa = pa;
  f.ts = nullptr;
  f.out = out;
  f.const_t = 1;
  f.tn = tn;
  ProfScope ps(c, "nl_forward_kernel");
  NLC_HIP(c, launch_nl_forward(f, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_model_forward(nlc_ctx* c, const double* obs, const double* window, const double* ts, int64_t N,
                                 int B, double* out, void* ws) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->has_model) return fail(c, NLC_ERR_STATE, "nlc_set_model has not been called");
  if (c->md.ilt.algo == NLC_ILT_DEHOOG && (c->S < 3 || c->S > 33 || c->S % 2 == 0))
    return fail(c, NLC_ERR_UNSUPPORTED, "dehoog: ilt_reconstruction_terms must be odd, 3 .. 33 (2M+1 terms)");
  if (N < 0 || B < 1) return fail(c, NLC_ERR_BAD_SHAPE, "bad N or B");
  if (N == 0) return NLC_OK;
  if (!obs || !window || !ts || !out || !ws) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  double* pa = (double*)ws;
  {
    GruArgs a = c->gru;
    a.mode = 0;
    a.window = window;
    a.N = N;
    a.B = B;
    a.out = pa;
    ProfScope ps(c, "gru_encode_kernel");
    NLC_HIP(c, launch_gru_encode(a, c->g, c->stream, gru_use_coop(c, a.N)));
  }
  if (c->md.ilt.algo == NLC_ILT_FIXED_TALBOT || c->md.ilt.algo == NLC_ILT_STEHFEST) {
    // staged (round 3): per-row query points s_k = node_k / t on the algorithm's own contour -> sphere inputs, the
    // representation kernel on explicit sphere inputs -> (theta, phi) rows, the Fourier kernel's stream with the algorithm's
    // per-term phase and weight
    const int S = c->S, dd = c->md.d;
    double* sph = pa + (N * 2 + 63) / 64 * 64;
    double* th = sph + (N * 2 * S + 63) / 64 * 64;
    double* ph = th + N * dd * S;
    const double* tab = nullptr;
    if (int r = linear_tables(c, &c->md.ilt, &tab)) return r;
    {
      RepInArgs ra{ts, ts, sph, N, 1, 0, S, 1, c->md.ilt.alpha, std::log(c->md.ilt.tol), c->md.ilt.scale, tab, tab + S, c->md.time_div};
      ProfScope ps(c, "rep_inputs_kernel");
      NLC_HIP(c, launch_rep_inputs(ra, c->stream));
    }
    RepFuncArgs rf{};
    rf.net = c->net;
    rf.N = N;
    rf.obs = obs;
    rf.obs_stride = dd;
    rf.obs_per_sample = 1;
    rf.Kep = 1;
    rf.pa = pa;
    rf.pa_stride = 2;
    rf.general_t = 1;
    rf.slot = c->slot_dev;
    rf.fre = th;
    rf.fim = ph;
    rf.sph = sph;
    rf.sph_stride = 2 * S;
    rf.write_angles = 1;
    {
      ProfScope ps(c, "nl_repfunc_kernel");
      NLC_HIP(c, launch_nl_repfunc(rf, c->stream));
    }
    IltArgs ia{th, ph, ts, out, N, dd, S, c->md.ilt.alpha, std::log(c->md.ilt.tol), c->md.ilt.scale, nullptr, nullptr, c->md.time_div, 1, 0, 0, 0};
    ia.lin_wr = tab + 2 * S;
    ia.lin_wi = tab + 3 * S;
    hipError_t le;
    {
      ProfScope ps(c, "ilt_linear_stream_kernel");
      le = launch_ilt_fourier(ia, c->stream);
    }
    if (le == hipErrorInvalidValue) {
      (void)hipGetLastError();
      return fail(c, NLC_ERR_UNSUPPORTED, "nlc_model_forward: this term count does not fit the stream kernel's tiling");
    }
    NLC_HIP(c, le);
    return NLC_OK;
  }
  if (c->md.ilt.algo == NLC_ILT_DEHOOG) {
    // staged: representation function -> F_k (re, im) in HBM -> de Hoog kernel (nonlinear in F: not an MFMA epilogue)
    double* fre = pa + (N * 2 + 63) / 64 * 64;
    double* fim = fre + N * c->md.d * c->S;
    RepFuncArgs rf{};
    rf.net = c->net;
    rf.N = N;
    rf.obs = obs;
    rf.obs_stride = c->md.d;
    rf.obs_per_sample = 1;
    rf.pa = pa;
    rf.pa_stride = 2;
    rf.ts = ts;
    rf.general_t = 1;
    rf.slot = c->slot_dev;
    rf.fre = fre;
    rf.fim = fim;
    {
      ProfScope ps(c, "nl_repfunc_kernel");
      NLC_HIP(c, launch_nl_repfunc(rf, c->stream));
    }
    IltArgs ia{nullptr, nullptr, ts, out, N, c->md.d, c->S, c->md.ilt.alpha, std::log(c->md.ilt.tol), c->md.ilt.scale,
               fre, fim, c->md.time_div, 1, 0, 0, 0};
    ProfScope ps(c, "ilt_dehoog_kernel");
    NLC_HIP(c, launch_ilt_dehoog(ia, c->stream));
    return NLC_OK;
  }
  {
    ForwardArgs f{};
    f.net = c->net;
    f.N = N;
    f.obs = obs;
    f.pa = pa;
    f.ts = ts;
    f.out = out;
    ProfScope ps(c, "nl_forward_kernel");
    NLC_HIP(c, launch_nl_forward(f, c->stream));
  }
  return NLC_OK;
  NLC_GUARD_END(c)
}

// LaplaceRepresentationFunc.forward (w_nl.py:55-63) on explicit input rows [theta_s (S) | phi_s (S) | p (d+2)]
extern "C" int nlc_rep_func(nlc_ctx* c, const double* rep_in, int64_t N, double* theta, double* phi) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->has_model) return fail(c, NLC_ERR_STATE, "nlc_set_model has not been called");
  if (N < 0) return fail(c, NLC_ERR_BAD_SHAPE, "bad N");
  if (N == 0) return NLC_OK;
  if (!rep_in || !theta || !phi) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  const int64_t row = 2 * (int64_t)c->S + c->P;
  RepFuncArgs rf{};
  rf.net = c->net;
  for (int i = 0; i < NLC_MAX_D; ++i) {  // the rows hold the latent p as the module sees it: no normalisation
    rf.net.state_mean[i] = 0.0;
    rf.net.state_std[i] = 1.0;
  }
  rf.N = N;
  rf.obs = rep_in + 2 * c->S;
  rf.obs_stride = row;
  rf.obs_per_sample = 1;
  rf.Kep = 1;
  rf.pa = rep_in + 2 * c->S + c->md.d;
  rf.pa_stride = row;
  rf.general_t = 1;
  rf.slot = c->slot_dev;
  rf.fre = theta;
  rf.fim = phi;
  rf.sph = rep_in;
  rf.sph_stride = row;
  rf.write_angles = 1;
  ProfScope ps(c, "nl_repfunc_kernel");
  NLC_HIP(c, launch_nl_repfunc(rf, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}

// =================================================================================== env side of the loop
extern "C" int nlc_env_step(nlc_ctx* c, int env, int friction, double dt, int delay, int64_t E, int B, int nu,
                            double* state, double* action_buffer, const double* action, double* obs, double* reward) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  static const int env_nu[3] = {1, 1, 2};
  if (env < 0 || env > 2) return fail(c, NLC_ERR_UNSUPPORTED, "unknown env id");
  if (nu != env_nu[env]) return fail(c, NLC_ERR_BAD_SHAPE, "nu does not match the env's action space");
  if (E < 0 || B < 1 || delay < 0 || delay > B - 1) return fail(c, NLC_ERR_BAD_SHAPE, "bad E / B / delay");
  if (E == 0) return NLC_OK;
  if (!state || !action_buffer || !action || !obs) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  EnvStepArgs a{env, friction, B, nu, delay, E, dt, state, action_buffer, action, obs, reward};
  ProfScope ps(c, "env_step_kernel");
  NLC_HIP(c, launch_env_step(a, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_env_obs(nlc_ctx* c, int env, int64_t E, const double* state, double* obs) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (env < 0 || env > 2) return fail(c, NLC_ERR_UNSUPPORTED, "unknown env id");
  if (E < 0) return fail(c, NLC_ERR_BAD_SHAPE, "bad E");
  if (E == 0) return NLC_OK;
  if (!state || !obs) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  EnvStepArgs a{env, 0, 1, 1, 0, E, 0.0, const_cast<double*>(state), nullptr, nullptr, obs, nullptr};
  ProfScope ps(c, "env_step_kernel");
  NLC_HIP(c, launch_env_step(a, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}

// =================================================================================== Delta-t RNN baseline
static int64_t rnn_blob_size(const nlc_rnn_desc* d) {
  const int64_t H = d->hidden;
  return 3 * H * d->nin + 3 * H * H + 6 * H + (int64_t)d->d * (H + d->d + (d->time_input ? 1 : 0)) + d->d;
}
extern "C" int64_t nlc_rnn_blob_size(const nlc_rnn_desc* d) { return d ? rnn_blob_size(d) : -1; }

extern "C" int nlc_set_rnn_model(nlc_ctx* c, const nlc_rnn_desc* d, const double* w, int64_t n) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!d || !w) return fail(c, NLC_ERR_BAD_ARG, "NULL desc or weights");
  if (d->hidden != 64 && d->hidden != 128 && d->hidden != 160)
    return fail(c, NLC_ERR_UNSUPPORTED, "DeltaTRNN hidden_units must be 64, 128 or 160");
  if (d->d < 1 || d->d > NLC_MAX_D) return fail(c, NLC_ERR_UNSUPPORTED, "state_dim must be in 1..8");
  if (d->nin < 1 || d->nin > NLC_MAX_NIN) return fail(c, NLC_ERR_UNSUPPORTED, "GRU input dim must be in 1..3");
  if (!(d->time_div != 0.0)) return fail(c, NLC_ERR_BAD_ARG, "time_div must be non-zero");
  if (n != rnn_blob_size(d)) return fail(c, NLC_ERR_BAD_SHAPE, "weight blob size mismatch");
  NLC_HIP(c, hipSetDevice(c->device));
  const int H = d->hidden, dd = d->d, nin = d->nin, F = H + dd + (d->time_input ? 1 : 0);
  Blob b{w, n};
  const double* Wih = b.take((int64_t)3 * H * nin);
  const double* Whh = b.take((int64_t)3 * H * H);
  const double* bih = b.take(3 * H);
  const double* bhh = b.take(3 * H);
  const double* Wo = b.take((int64_t)dd * F);
  const double* bo = b.take(dd);
  DeviceArena ar;
  // input weights with the biases folded into input column 3 (x = [a_0..a_{nin-1}, 0.., 1]), as for the NL encoder
  std::vector<double> Wihb((size_t)3 * H * 4, 0.0);
  for (int r = 0; r < 3 * H; ++r) {
    for (int j = 0; j < nin; ++j) Wihb[(size_t)r * 4 + j] = Wih[(size_t)r * nin + j];
    Wihb[(size_t)r * 4 + 3] = bih[r] + (r < 2 * H ? bhh[r] : 0.0);
  }
  const size_t o_Wih = ar.push(pack_gru_chunked(Wihb.data(), 4, 4, H));
  const size_t o_Whh = ar.push(pack_gru_chunked(Whh, H, H, H));
  const size_t o_bhn = ar.push(std::vector<double>(bhh + 2 * H, bhh + 3 * H));
  const size_t o_Wo = ar.push(pack_A(Wo, F, H, identity_rows(dd)));
  double* base = nullptr;
  NLC_HIP(c, hipMalloc((void**)&base, ar.host.size() * sizeof(double)));
  hipError_t e = hipMemcpy(base, ar.host.data(), ar.host.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    hipFree(base);
    return fail(c, NLC_ERR_HIP, std::string("weight upload: ") + hipGetErrorString(e));
  }
  NLC_HIP(c, hipStreamSynchronize(c->stream));
  if (c->rnn_base) hipFree(c->rnn_base);
  c->rnn_base = base;
  c->rd = *d;
  RnnArgs& R = c->rnn;
  R = RnnArgs{};
  R.nin = nin;
  R.d = dd;
  for (int j = 0; j < nin; ++j) {
    R.mean[j] = d->action_mean[j];
    R.std[j] = d->action_std[j];
  }
  R.Wihp = base + o_Wih;
  R.Whhp = base + o_Whh;
  R.bhn = base + o_bhn;
  R.Wop = base + o_Wo;
  RnnHead& Hd = c->rnn_head;
  Hd = RnnHead{};
  Hd.d = dd;
  for (int i = 0; i < dd; ++i) {
    for (int j = 0; j < dd; ++j) Hd.Wx[i * dd + j] = Wo[(size_t)i * F + H + j];
    Hd.wt[i] = d->time_input ? Wo[(size_t)i * F + H + dd] : 0.0;
    Hd.b[i] = bo[i];
    Hd.mean[i] = d->state_mean[i];
    Hd.std[i] = d->state_std[i];
  }
  Hd.time_div = d->time_div;
  c->has_rnn = true;
  if (c->has_mppi && c->pd.dynamics == NLC_DYN_DTRNN) c->has_mppi = false;  // re-configure against the new weights
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_rnn_forward(nlc_ctx* c, const double* obs, const double* window, const double* ts, int64_t N, int B,
                               double* out, void* ws) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->has_rnn) return fail(c, NLC_ERR_STATE, "nlc_set_rnn_model has not been called");
  if (N < 0 || B < 1) return fail(c, NLC_ERR_BAD_SHAPE, "bad N or B");
  if (N == 0) return NLC_OK;
  if (!obs || !window || !ts || !out || !ws) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  RnnArgs a = c->rnn;
  a.mode = 0;
  a.window = window;
  a.N = N;
  a.B = B;
  a.out = (double*)ws;
  {
    ProfScope ps(c, "rnn_encode_kernel");
    NLC_HIP(c, launch_rnn_encode(a, c->rd.hidden, c->stream));
  }
  RnnForwardArgs f{};
  f.head = c->rnn_head;
  f.N = N;
  f.obs = obs;
  f.q = (const double*)ws;
  f.ts = ts;
  f.out = out;
  ProfScope ps(c, "rnn_forward_tail_kernel");
  NLC_HIP(c, launch_rnn_forward_tail(f, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}

// =================================================================================== planner
// =================================================================================== NODE baseline
namespace {
int64_t node_blob_size(const nlc_node_desc* d) {
  const int64_t H = d->hidden, dy = d->d + d->augment_dim, in = dy + d->nu;
  return H * in + H + H * H + H + dy * H + dy;
}
// step sizes of torchdiffeq's fixed-grid solver over [0, t_end] (restated: oracle/node_model.py::euler_substeps)
int node_substeps(double t_end, double step, double* h, int max_n) {
  const int niters = (int)std::ceil(t_end / step + 1.0);
  if (niters < 2 || niters - 1 > max_n) return -1;
  double prev = 0.0;
  for (int i = 1; i < niters; ++i) {
    const double tk = (i == niters - 1) ? t_end : (double)i * step;
    h[i - 1] = tk - prev;
    prev = tk;
  }
  return niters - 1;
}
}  // namespace
extern "C" int64_t nlc_node_blob_size(const nlc_node_desc* d) { return d ? node_blob_size(d) : -1; }

extern "C" int nlc_set_node_model(nlc_ctx* c, const nlc_node_desc* d, const double* w, int64_t n) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!d || !w) return fail(c, NLC_ERR_BAD_ARG, "NULL desc or weights");
  if (d->hidden < 1 || d->hidden > 272) return fail(c, NLC_ERR_UNSUPPORTED, "NODE hidden_units must be in 1..272");
  if (d->d < 1 || d->augment_dim < 0 || d->d + d->augment_dim > 8 || d->d > NLC_MAX_D)
    return fail(c, NLC_ERR_UNSUPPORTED, "state_dim + augment_dim must be <= 8");
  if (d->nu < 1 || d->nu > NLC_MAX_NU) return fail(c, NLC_ERR_UNSUPPORTED, "nu must be 1 or 2");
  if (!(d->time_div != 0.0) || !(d->step_size > 0.0)) return fail(c, NLC_ERR_BAD_ARG, "bad time_div / step_size");
  if (n != node_blob_size(d)) return fail(c, NLC_ERR_BAD_SHAPE, "weight blob size mismatch");
  NLC_HIP(c, hipSetDevice(c->device));
  const int H = d->hidden, dy = d->d + d->augment_dim, in = dy + d->nu;
  const int ht = H <= 64 ? 4 : (H <= 128 ? 8 : 17), Hp = 16 * ht;
  Blob b{w, n};
  const double* W1 = b.take((int64_t)H * in);
  const double* b1 = b.take(H);
  const double* W2 = b.take((int64_t)H * H);
  const double* b2 = b.take(H);
  const double* W3 = b.take((int64_t)dy * H);
  const double* b3 = b.take(dy);
  // pad the hidden width to Hp rows / columns with zeros; inputs to 12 columns
  std::vector<double> W1z((size_t)Hp * 12, 0.0), W2z((size_t)Hp * Hp, 0.0), W3z((size_t)16 * Hp, 0.0), b1z(Hp, 0.0),
      b2z(Hp, 0.0), b3z(16, 0.0);
  for (int r = 0; r < H; ++r) {
    for (int j = 0; j < in; ++j) W1z[(size_t)r * 12 + j] = W1[(size_t)r * in + j];
    for (int j = 0; j < H; ++j) W2z[(size_t)r * Hp + j] = W2[(size_t)r * H + j];
    b1z[r] = b1[r];
    b2z[r] = b2[r];
  }
  for (int r = 0; r < dy; ++r) {
    for (int j = 0; j < H; ++j) W3z[(size_t)r * Hp + j] = W3[(size_t)r * H + j];
    b3z[r] = b3[r];
  }
  DeviceArena ar;
  const size_t o_W1 = ar.push(pack_A(W1z.data(), 12, 12, identity_rows(Hp)));
  const size_t o_b1 = ar.push(b1z);
  const size_t o_W2 = ar.push(pack_A(W2z.data(), Hp, Hp, identity_rows(Hp)));
  const size_t o_b2 = ar.push(b2z);
  const size_t o_W3 = ar.push(pack_A(W3z.data(), Hp, Hp, identity_rows(16)));
  const size_t o_b3 = ar.push(b3z);
  double* base = nullptr;
  NLC_HIP(c, hipMalloc((void**)&base, ar.host.size() * sizeof(double)));
  hipError_t e = hipMemcpy(base, ar.host.data(), ar.host.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    hipFree(base);
    return fail(c, NLC_ERR_HIP, std::string("weight upload: ") + hipGetErrorString(e));
  }
  NLC_HIP(c, hipStreamSynchronize(c->stream));
  if (c->node_base) hipFree(c->node_base);
  c->node_base = base;
  c->nd = *d;
  c->node_ht = ht;
  NodeNetArgs& N = c->node;
  N = NodeNetArgs{};
  N.d = d->d;
  N.aug = d->augment_dim;
  N.nu = d->nu;
  N.W1p = base + o_W1;
  N.b1 = base + o_b1;
  N.W2p = base + o_W2;
  N.b2 = base + o_b2;
  N.W3p = base + o_W3;
  N.b3 = base + o_b3;
  for (int i = 0; i < NLC_MAX_D; ++i) {
    N.state_mean[i] = i < d->d ? d->state_mean[i] : 0.0;
    N.state_std[i] = i < d->d ? d->state_std[i] : 1.0;
  }
  c->has_node = true;
  if (c->has_mppi && c->pd.dynamics == NLC_DYN_NODE) c->has_mppi = false;
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_node_forward(nlc_ctx* c, const double* obs, const double* action, double ts_pred, int64_t N,
                                double* out) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->has_node) return fail(c, NLC_ERR_STATE, "nlc_set_node_model has not been called");
  if (N < 0) return fail(c, NLC_ERR_BAD_SHAPE, "bad N");
  if (N == 0) return NLC_OK;
  if (!obs || !action || !out) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  NodeForwardArgs f{};
  f.net = c->node;
  f.net.nsub = node_substeps(ts_pred / c->nd.time_div, c->nd.step_size, f.net.hsub, 8);
  if (f.net.nsub < 0) return fail(c, NLC_ERR_UNSUPPORTED, "prediction time needs more than 8 Euler sub-steps (or is <= 0)");
  f.N = N;
  f.obs = obs;
  f.action = action;
  f.out = out;
  ProfScope ps(c, "node_forward_kernel");
  NLC_HIP(c, launch_node_forward(f, c->node_ht, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_mppi_configure(nlc_ctx* c, const nlc_mppi_desc* d) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!d) return fail(c, NLC_ERR_BAD_ARG, "NULL desc");
  if (d->K < 1 || d->T < 1 || d->K_global < d->K || d->k_offset < 0 || d->k_offset + d->K > d->K_global)
    return fail(c, NLC_ERR_BAD_SHAPE, "bad K / K_global / k_offset / T");
  if (d->nu < 1 || d->nu > NLC_MAX_NU) return fail(c, NLC_ERR_UNSUPPORTED, "nu must be 1 or 2");
  if (d->d < 1 || d->d > NLC_MAX_D) return fail(c, NLC_ERR_BAD_SHAPE, "bad nx");
  if (d->B < 1) return fail(c, NLC_ERR_BAD_SHAPE, "action_buffer needs at least one row");
  if (!(d->lambda_ > 0.0) || d->u_scale == 0.0) return fail(c, NLC_ERR_BAD_ARG, "lambda_ must be > 0, u_scale != 0");
  if (d->u_per_command < 1 || d->u_per_command > d->T) return fail(c, NLC_ERR_BAD_ARG, "bad u_per_command");
  if (d->E < 0 || d->E > 65535) return fail(c, NLC_ERR_BAD_ARG, "episodes E must be in [0, 65535]");
  const int E = d->E < 1 ? 1 : d->E;
  if ((double)E * (double)d->K * d->T * d->nu > 2.0e9)
    return fail(c, NLC_ERR_BAD_SHAPE, "E*K*T*nu exceeds the planner's index range");
  // the env id selects the running cost and the oracle dynamics; with cost_external and NL dynamics nothing needs it
  const bool env_free = d->cost_external &&
                        (d->dynamics == NLC_DYN_NL || d->dynamics == NLC_DYN_DTRNN || d->dynamics == NLC_DYN_NODE) &&
                        d->env == -1;
  if (!env_free && (d->env < 0 || d->env > 2)) return fail(c, NLC_ERR_UNSUPPORTED, "unknown env id");
  static const int env_d[3] = {5, 3, 6}, env_nu[3] = {1, 1, 2};
  if (!env_free && d->dynamics != NLC_DYN_EXTERNAL && (d->d != env_d[d->env] || d->nu != env_nu[d->env]))
    return fail(c, NLC_ERR_BAD_SHAPE, "nx / nu do not match the env's trig observation");
  if (d->cost_external && d->dynamics == NLC_DYN_EXTERNAL)
    return fail(c, NLC_ERR_BAD_ARG, "cost_external needs fused dynamics (NLC_DYN_NL / NLC_DYN_ORACLE / NLC_DYN_DTRNN)");
  if (d->dynamics == NLC_DYN_EXTERNAL) {
    // the caller owns dynamics and cost
  } else if (d->dynamics == NLC_DYN_NL) {
    if (!c->has_model) return fail(c, NLC_ERR_STATE, "NL dynamics need nlc_set_model first");
    // nin == nu + 1: an encode_obs_time model; the rollout appends the harness's constant time channel
    if (c->md.d != d->d || (c->md.nin != d->nu && c->md.nin != d->nu + 1))
      return fail(c, NLC_ERR_BAD_SHAPE, "model state/action dims differ from the planner's");
    if (c->md.ilt.algo == NLC_ILT_DEHOOG && (c->S < 3 || c->S > 33 || c->S % 2 == 0))
      return fail(c, NLC_ERR_UNSUPPORTED, "dehoog: ilt_reconstruction_terms must be odd, 3 .. 33 (2M+1 terms)");
  } else if (d->dynamics == NLC_DYN_ORACLE) {
    if (d->delay < 0 || d->delay > d->B - 1)
      return fail(c, NLC_ERR_BAD_ARG, "oracle dynamics: delay must be in [0, action_buffer_size-1]");
  } else if (d->dynamics == NLC_DYN_DTRNN) {
    if (!c->has_rnn) return fail(c, NLC_ERR_STATE, "Delta-t RNN dynamics need nlc_set_rnn_model first");
    if (c->rd.d != d->d || c->rd.nin != d->nu)
      return fail(c, NLC_ERR_BAD_SHAPE, "model state/action dims differ from the planner's");
  } else if (d->dynamics == NLC_DYN_NODE) {
    if (!c->has_node) return fail(c, NLC_ERR_STATE, "NODE dynamics need nlc_set_node_model first");
    if (c->nd.d != d->d || c->nd.nu != d->nu)
      return fail(c, NLC_ERR_BAD_SHAPE, "model state/action dims differ from the planner's");
    double h[8];
    if (node_substeps(d->ts_pred / c->nd.time_div, c->nd.step_size, h, 8) < 0)
      return fail(c, NLC_ERR_UNSUPPORTED, "ts_pred needs more than 8 Euler sub-steps (or is <= 0)");
  } else {
    return fail(c, NLC_ERR_UNSUPPORTED, "unknown dynamics id");
  }
  NLC_HIP(c, hipSetDevice(c->device));
  const size_t un = (size_t)E * d->T * d->nu;
  for (int i = 0; i < 2; ++i) {
    if (c->U[i]) hipFree(c->U[i]);
    c->U[i] = nullptr;
    NLC_HIP(c, hipMalloc((void**)&c->U[i], un * sizeof(double)));
    // on the ctx's stream and waited for below: hipMemset runs asynchronously on the NULL stream, which is not
    // ordered against a non-blocking stream -- it could land after the nlc_mppi_set_U copy that follows configure
    NLC_HIP(c, hipMemsetAsync(c->U[i], 0, un * sizeof(double), c->stream));
  }
  NLC_HIP(c, hipStreamSynchronize(c->stream));
  c->ucur = 0;
  if (c->small) hipFree(c->small);
  c->small = nullptr;
  NLC_HIP(c, hipMalloc((void**)&c->small, (un + 2 * (size_t)E) * sizeof(double)));
  const size_t pin_need = (size_t)E * d->d + (size_t)E * d->B * d->nu + un + 8;
  if (pin_need > c->pinned_n) {
    if (c->pinned) hipHostFree(c->pinned);
    c->pinned = nullptr;
    NLC_HIP(c, hipHostMalloc((void**)&c->pinned, pin_need * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    c->pinned_n = pin_need;
  }
  std::memset(c->pinned, 0, c->pinned_n * sizeof(double));
  c->pd = *d;
  c->pd.E = E;
  c->nblk = weight_tiles(d->K);
  if (d->dynamics == NLC_DYN_NL) {
    // constant prediction time => the 2S sphere-coordinate inputs of layer 1 are constants: fold into the bias
    c->tn = d->ts_pred / c->md.time_div;
    std::vector<double> sph;
    sphere_inputs(c->md.ilt, c->tn, sph);
    const int h = c->md.h, S = c->S;
    std::vector<double> bf(h);
    for (int r = 0; r < h; ++r) {
      double acc = c->b1_host[r];
      for (int j = 0; j < 2 * S; ++j) acc += c->W1s_host[(size_t)r * 2 * S + j] * sph[j];
      bf[r] = acc;
    }
    if (c->b1fold) hipFree(c->b1fold);
    c->b1fold = nullptr;
    NLC_HIP(c, hipMalloc((void**)&c->b1fold, h * sizeof(double)));
    NLC_HIP(c, hipMemcpy(c->b1fold, bf.data(), h * sizeof(double), hipMemcpyHostToDevice));
    if (c->md.ilt.algo == NLC_ILT_FIXED_TALBOT || c->md.ilt.algo == NLC_ILT_STEHFEST) {
      // coefficient fragments of the LIN rollout instances: lane -> (dim = lane & 15, slot 4 g + (lane >> 4)), as Cp
      std::vector<double> tab;
      linear_tables_host(c->md.ilt.algo, S, tab);
      const int ng = 2 * c->net.nt3;
      std::vector<double> cp((size_t)2 * ng * 64, 0.0);
      for (int g = 0; g < ng; ++g)
        for (int lane = 0; lane < 64; ++lane) {
          const auto el = c->slot_elems[(size_t)4 * g + (lane >> 4)];
          if (el.first != (lane & 15)) continue;
          cp[(size_t)g * 64 + lane] = tab[2 * S + el.second] / c->tn;
          cp[(size_t)(ng + g) * 64 + lane] = -tab[3 * S + el.second] / c->tn;
        }
      if (c->cp_lin) hipFree(c->cp_lin);
      c->cp_lin = nullptr;
      NLC_HIP(c, hipMalloc((void**)&c->cp_lin, cp.size() * sizeof(double)));
      NLC_HIP(c, hipMemcpy(c->cp_lin, cp.data(), cp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  c->has_mppi = true;
  c->sync_clean_ws = nullptr;
  c->sync_dirty = false;
  c->last.valid = false;
  return NLC_OK;
  NLC_GUARD_END(c)
}

namespace {
// fixed Talbot / Stehfest models whose rollout runs on the LIN instances of the rollout kernels (kernels_nl_lin*.hip) instead of
// the staged path
bool linear_on_rollout_kernels(const nlc_ctx* c) {
  return c->has_model && (c->md.ilt.algo == NLC_ILT_FIXED_TALBOT || c->md.ilt.algo == NLC_ILT_STEHFEST) &&
         (c->md.h == 64 || c->md.h == 128 || c->md.h == 256) && c->opt_linear_fused != 0;
}

struct WsLayout {
  size_t tile_part, chunk_part, pa, state0, abuf, xcarry, ccarry, fre, fim, dx, tconst, rq, sync, total;
};
WsLayout ws_layout(const nlc_ctx* c) {
  const nlc_mppi_desc& d = c->pd;
  WsLayout w{};
  size_t off = 0;
  auto take = [&](size_t n) {
    const size_t o = off;
    off += (n + 63) / 64 * 64;
    return o;
  };
  const size_t KE = (size_t)d.K * d.E;  // all local samples
  w.tile_part = take((size_t)d.E * c->nblk * (2 + (size_t)d.T * d.nu));
  w.chunk_part = take((size_t)d.E * ((c->nblk + 63) / 64) * (2 + (size_t)d.T * d.nu));
  w.pa = take(d.dynamics == NLC_DYN_NL ? KE * d.T * 2 : 0);
  w.state0 = take(d.dynamics == NLC_DYN_EXTERNAL ? 0 : KE * d.d);
  w.abuf = take((size_t)d.E * d.B * d.nu);
  w.xcarry = take(d.dynamics == NLC_DYN_NL ? KE * d.d : 0);
  w.ccarry = take(d.dynamics == NLC_DYN_NL ? KE * 2 : 0);
  // (the staged buffers are laid out for every non-Fourier model, also when a linear-algorithm model runs on the LIN rollout
  // instances: the layout must not depend on an option that can change after the caller sized its workspace)
  const bool staged = d.dynamics == NLC_DYN_NL && c->md.ilt.algo != NLC_ILT_FOURIER;
  w.fre = take(staged ? KE * 8 * (size_t)c->net.nt3 : 0);  // slot-major (8*nt3, KE), >= KE*d*S
  w.fim = take(staged ? KE * 8 * (size_t)c->net.nt3 : 0);
  w.dx = take(staged ? KE * d.d : 0);
  w.tconst = take(staged ? 8 : 0);
  w.rq = take(d.dynamics == NLC_DYN_DTRNN ? KE * d.T * d.d : 0);  // hidden part of linear_out, (T, K, d)
  // fused one-launch planner body: tickets, per-CU census, one flag word per encoder tile (unsigned words)
  w.sync = take(d.dynamics == NLC_DYN_NL && !staged ? (fused_sync_words(d.T, (int64_t)KE) + 1) / 2 : 0);
  w.total = off;
  return w;
}
}  // namespace

// pinned host word a rollout workgroup of the fused planner body sets when it gives up waiting for an encoder tile
static double* fused_timeout_word(nlc_ctx* c) {
  const nlc_mppi_desc& d = c->pd;
  return c->pinned + (size_t)d.E * d.d + (size_t)d.E * d.B * d.nu + (size_t)d.E * d.T * d.nu;
}
static bool fused_gave_up(nlc_ctx* c) {
  unsigned* w = reinterpret_cast<unsigned*>(fused_timeout_word(c));
  if (*w == 0u) return false;
  *w = 0u;
  return true;
}

static WeightArgs make_weight_args(nlc_ctx* c, const nlc_mppi_buffers* buf) {
  const nlc_mppi_desc& d = c->pd;
  const WsLayout w = ws_layout(c);
  double* ws = (double*)buf->workspace;
  WeightArgs wa{};
  wa.Kep = d.K;
  wa.E = d.E;
  wa.T = d.T;
  wa.nu = d.nu;
  wa.lambda_ = d.lambda_;
  wa.cost = buf->cost_total;
  wa.noise = buf->noise;
  wa.tile_part = ws + w.tile_part;
  wa.chunk_part = ws + w.chunk_part;
  wa.partials = buf->partials;
  wa.nblk = c->nblk;
  return wa;
}

static int run_weights(nlc_ctx* c, const nlc_mppi_buffers* buf) {
  const WeightArgs wa = make_weight_args(c, buf);
  ProfScope ps(c, "weight_kernels");
  NLC_HIP(c, launch_weights(wa, c->stream));
  return NLC_OK;
}

extern "C" int64_t nlc_mppi_workspace_bytes(nlc_ctx* c) {
  if (!c || !c->has_mppi) return -1;
  return (int64_t)(ws_layout(c).total * sizeof(double));
}

extern "C" int nlc_mppi_set_U(nlc_ctx* c, const double* U) {
  if (!c) return NLC_ERR_BAD_ARG;
  if (!c->has_mppi) return fail(c, NLC_ERR_STATE, "planner not configured");
  if (!U) return fail(c, NLC_ERR_BAD_ARG, "NULL U");
  NLC_HIP(c, hipSetDevice(c->device));
  NLC_HIP(c, hipMemcpyAsync(c->U[c->ucur], U, (size_t)c->pd.E * c->pd.T * c->pd.nu * sizeof(double), hipMemcpyHostToDevice,
                            c->stream));
  NLC_HIP(c, hipStreamSynchronize(c->stream));
  return NLC_OK;
}

extern "C" int nlc_mppi_get_U(nlc_ctx* c, double* U) {
  if (!c) return NLC_ERR_BAD_ARG;
  if (!c->has_mppi) return fail(c, NLC_ERR_STATE, "planner not configured");
  if (!U) return fail(c, NLC_ERR_BAD_ARG, "NULL U");
  NLC_HIP(c, hipSetDevice(c->device));
  NLC_HIP(c, hipMemcpyAsync(U, c->U[c->ucur], (size_t)c->pd.E * c->pd.T * c->pd.nu * sizeof(double), hipMemcpyDeviceToHost,
                            c->stream));
  NLC_HIP(c, hipStreamSynchronize(c->stream));
  return NLC_OK;
}

// Phase 1 of a command.  `replay`: the command's fused launch gave up (hand-off timeout) -- run it again on the two-launch
// body from the inputs kept in c->last / still staged in the workspace (nlc_mppi_finish).
static int mppi_rollout_impl(nlc_ctx* c, const double* state, int state_per_sample, const double* abuf_host,
                             const nlc_mppi_buffers* buf, int rng, uint64_t seed, uint64_t counter, bool replay) {
  NLC_GUARD_BEGIN
  if (!c->has_mppi) return fail(c, NLC_ERR_STATE, "planner not configured");
  const nlc_mppi_desc& d = c->pd;
  const bool external = d.dynamics == NLC_DYN_EXTERNAL;
  if ((!external && (!state || !abuf_host)) || !buf) return fail(c, NLC_ERR_BAD_ARG, "NULL argument");
  if (!buf->noise || !buf->perturbed || !buf->cost_total || !buf->cost_nz || !buf->partials || !buf->workspace)
    return fail(c, NLC_ERR_BAD_ARG, "NULL required device buffer");
  if (d.cost_external && !buf->states) return fail(c, NLC_ERR_BAD_ARG, "cost_external needs buf->states");
  NLC_HIP(c, hipSetDevice(c->device));
  if (!replay && fused_gave_up(c)) {  // (a device-resident caller never synchronised inside nlc_mppi_finish)
    c->fused_lost = true;  // from here on the two-launch body
    return fail(c, NLC_ERR_HIP, "fused planner body: the PREVIOUS command timed out waiting inside the launch (its action "
                                "was not valid); later commands run the two-launch body");
  }
  const WsLayout w = ws_layout(c);
  double* ws = (double*)buf->workspace;
  double* state_dev = ws + w.state0;
  double* abuf_dev = ws + w.abuf;
  const int64_t KE = d.K * d.E;  // all local samples, episode-major
  bool inline_inputs = false;
  if (replay) {
    inline_inputs = c->last.inline_inputs;  // otherwise the inputs are still staged in the workspace
  } else if (!external && d.E > 1) {
    // batched episodes: the (E,d) states and (E,B,nu) action buffers usually live on the device already (a device-side
    // env loop); hipMemcpyDefault takes either kind of pointer
    const size_t ns = (size_t)(state_per_sample ? KE : d.E) * d.d;
    NLC_HIP(c, hipMemcpyAsync(state_dev, state, ns * sizeof(double), hipMemcpyDefault, c->stream));
    NLC_HIP(c, hipMemcpyAsync(abuf_dev, abuf_host, (size_t)d.E * d.B * d.nu * sizeof(double), hipMemcpyDefault,
                              c->stream));
    // a caller-owned HOST buffer may be a temporary: the copies out of it must have completed before we return
    if (!is_device_ptr(state) || !is_device_ptr(abuf_host)) NLC_HIP(c, hipStreamSynchronize(c->stream));
  } else if (!external && !state_per_sample && d.B * d.nu <= kMaxInlineAbuf) {
    inline_inputs = true;  // state and action_buffer travel in the shift kernel's arguments (below)
  } else if (!external) {
    // small inputs go through pinned staging (truly asynchronous copies).  The staging slots are rewritten
    // only after the previous command's copies out of them have completed (stage_ev).
    if (c->stage_ev) NLC_HIP(c, hipEventSynchronize(c->stage_ev));
    double* pin_state = c->pinned;
    double* pin_abuf = c->pinned + d.d;
    std::memcpy(pin_abuf, abuf_host, (size_t)d.B * d.nu * sizeof(double));
    if (state_per_sample) {
      // (K, d) from caller-owned pageable memory: wait for the copy, the source may be a temporary
      NLC_HIP(c, hipMemcpyAsync(state_dev, state, (size_t)d.K * d.d * sizeof(double), hipMemcpyHostToDevice, c->stream));
      NLC_HIP(c, hipStreamSynchronize(c->stream));
    } else {
      std::memcpy(pin_state, state, (size_t)d.d * sizeof(double));
      NLC_HIP(c, hipMemcpyAsync(state_dev, pin_state, (size_t)d.d * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    NLC_HIP(c, hipMemcpyAsync(abuf_dev, pin_abuf, (size_t)d.B * d.nu * sizeof(double), hipMemcpyHostToDevice,
                              c->stream));
    if (!c->stage_ev) NLC_HIP(c, hipEventCreateWithFlags(&c->stage_ev, hipEventDisableTiming));
    NLC_HIP(c, hipEventRecord(c->stage_ev, c->stream));
  }
  PerturbArgs p{};
  p.K = KE;
  p.Kep = d.K;
  p.E = d.E;
  p.K_global = d.K_global;
  p.k_offset = d.k_offset;
  p.T = d.T;
  p.nu = d.nu;
  p.U_old = c->U[c->ucur];
  p.U_new = c->U[c->ucur ^ 1];
  p.noise = buf->noise;
  p.perturbed = buf->perturbed;
  p.actions = buf->actions;
  p.u_scale = d.u_scale;
  p.has_bounds = d.has_bounds;
  p.sample_null_action = d.sample_null_action;
  p.rng = rng;
  for (int i = 0; i < NLC_MAX_NU; ++i) {
    p.u_min[i] = d.u_min[i];
    p.u_max[i] = d.u_max[i];
    p.u_init[i] = d.u_init[i];
    p.mu[i] = d.noise_mu[i];
  }
  for (int i = 0; i < NLC_MAX_NU * NLC_MAX_NU; ++i) p.chol[i] = d.noise_chol[i];
  p.seed = seed;
  p.counter = counter;
  if (inline_inputs) {
    p.n_state_in = d.d;
    p.n_abuf_in = d.B * d.nu;
    p.state_dst = state_dev;
    p.abuf_dst = abuf_dev;
    std::memcpy(p.state_in, state, (size_t)d.d * sizeof(double));
    std::memcpy(p.abuf_in, abuf_host, (size_t)d.B * d.nu * sizeof(double));
  }
  if (!replay) {
    nlc_ctx::LastCommand& L = c->last;
    L.valid = !external;
    L.inline_inputs = inline_inputs;
    L.fused = false;
    L.state_per_sample = state_per_sample;
    L.rng = rng;
    L.seed = seed;
    L.counter = counter;
    if (inline_inputs) {
      std::memcpy(L.state_in, p.state_in, sizeof(L.state_in));
      std::memcpy(L.abuf_in, p.abuf_in, sizeof(L.abuf_in));
    }
  }
  c->ucur ^= 1;  // from here on c->U[c->ucur] is the shifted sequence the two kernels below produce
  // (a lambda: the fused planner body's argument block rides to the device in the perturb kernel's arguments, so that
  // path fills p.blob first)
  auto launch_shift_perturb = [&]() -> int {
    // one launch: the perturb kernel shifts U on the fly (every (k, t) thread reads U_old[t + 1]; the episode's first
    // local sample stores the shifted row) and stores the staged inputs; shift_U_kernel only when there is nothing to
    // perturb
    p.fused_shift = (int64_t)p.K * p.T > 0;
    if (!p.fused_shift) {
      ProfScope ps(c, "shift_U_kernel");
      NLC_HIP(c, launch_shift_U(p, c->stream));
    }
    ProfScope ps(c, "perturb_kernel");
    NLC_HIP(c, launch_perturb(p, c->stream));
    return NLC_OK;
  };
  const bool lin_direct = d.dynamics == NLC_DYN_NL && linear_on_rollout_kernels(c);
  const bool nl_fourier = d.dynamics == NLC_DYN_NL && (c->md.ilt.algo == NLC_ILT_FOURIER || lin_direct);  // one rollout kernel
  if (!nl_fourier)
    if (int rc = launch_shift_perturb()) return rc;
  if (external) return NLC_OK;  // the caller runs the horizon loop, then nlc_mppi_weights
  if (d.dynamics == NLC_DYN_NL) {
    double* pa = ws + w.pa;
    GruArgs g = c->gru;
    g.mode = 1;
    g.perturbed = buf->perturbed;
    g.abuf = abuf_dev;
    g.u_scale = d.u_scale;
    g.T = d.T;
    g.B = d.B;
    g.Kep = d.K;
    g.nact = d.nu;
    g.out = pa;
    RolloutArgs r{};
    r.net = c->net;
    r.net.b1 = c->b1fold;
    r.K = KE;
    r.Kep = d.K;
    r.T = d.T;
    r.nu = d.nu;
    r.B = d.B;
    r.env = d.cost_external ? -1 : d.env;  // only the running cost reads it in the NL rollout
    r.state_per_sample = state_per_sample;
    r.state0 = state_dev;
    r.pa = pa;
    r.perturbed = buf->perturbed;
    r.noise = buf->noise;
    r.U = c->U[c->ucur];
    for (int i = 0; i < NLC_MAX_NU * NLC_MAX_NU; ++i) r.sigma_inv[i] = d.noise_sigma_inv[i];
    r.lambda_ = d.lambda_;
    r.u_scale = d.u_scale;
    r.noise_abs_cost = d.noise_abs_cost;
    r.tn = c->tn;
    r.states = buf->states;
    r.cost_total = buf->cost_total;
    r.xcarry = ws + w.xcarry;
    r.ccarry = ws + w.ccarry;
    if (lin_direct) {
      if (!c->cp_lin) return fail(c, NLC_ERR_STATE, "linear-algorithm coefficient tables missing (nlc_mppi_configure)");
      const size_t ng = (size_t)2 * c->net.nt3 * 64;
      r.net.Cp = c->cp_lin;
      r.net.Cp2 = c->cp_lin + ng;
      r.net.lin = 1;
    }
    if (c->md.ilt.algo != NLC_ILT_FOURIER && !lin_direct) {
      // staged de Hoog planner path (BASELINE configs[4]): hoisted GRU, then per horizon step three launches --
      // representation function -> F_k, de Hoog ILT -> dx, state/cost tail.  Everything stays on the device.
      // fixed_tablot / stehfest models (round 3) take the same path with the slot-major linear ILT in de Hoog's place.
      const bool dehoog = c->md.ilt.algo == NLC_ILT_DEHOOG;
      const double* lin_tab = nullptr;
      if (!dehoog)
        if (int rc = linear_tables(c, &c->md.ilt, &lin_tab)) return rc;
      // GRU encode: one launch up front, or (round 3, option "dehoog_gru_chunks" C > 1) C horizon chunks on a stream of
      // their own that run BESIDE the step chain -- the chain's launches wait for the chunk that holds their horizon step.
      // The chunks use the cooperative kernel at reduced occupancy (lds_pad) so that the chain's workgroups find room.
      int C = d.E == 1 ? c->opt_dehoog_gru_chunks : 1;
      if (C == 0) C = 1;  // auto: off
      if (C > d.T) C = d.T;
      if (C > 8) C = 8;
      const int Tc = (d.T + C - 1) / C;
      if (C == 1) {
        g.t0 = 0;
        g.Tc = d.T;
        g.N = KE * d.T;
        ProfScope ps(c, "gru_encode_kernel");
        NLC_HIP(c, launch_gru_encode(g, c->g, c->stream, gru_use_coop(c, g.N)));
      }
      double* tconst = ws + w.tconst;
      NLC_HIP(c, hipMemcpyAsync(tconst, &c->tn, sizeof(double), hipMemcpyHostToDevice, c->stream));
      if (C > 1) {
        if (!c->gru_stream) NLC_HIP(c, hipStreamCreateWithFlags(&c->gru_stream, hipStreamNonBlocking));
        if (!c->ev_fork) NLC_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        while ((int)c->ev_gru.size() < C) {
          hipEvent_t e2 = nullptr;
          NLC_HIP(c, hipEventCreateWithFlags(&e2, hipEventDisableTiming));
          c->ev_gru.push_back(e2);
        }
        NLC_HIP(c, hipEventRecord(c->ev_fork, c->stream));  // behind the perturb kernel and the staged inputs
        NLC_HIP(c, hipStreamWaitEvent(c->gru_stream, c->ev_fork, 0));
        for (int ch = 0; ch < C; ++ch) {
          g.t0 = ch * Tc;
          g.Tc = (g.t0 + Tc <= d.T) ? Tc : d.T - g.t0;
          if (g.Tc <= 0) break;
          g.N = KE * g.Tc;
          {
            ProfScope ps(c, "gru_encode_kernel", c->gru_stream, true);
            NLC_HIP(c, launch_gru_encode(g, c->g, c->gru_stream, true, (unsigned)c->opt_dehoog_gru_lds_pad));
          }
          NLC_HIP(c, hipEventRecord(c->ev_gru[ch], c->gru_stream));
        }
      }
      RepFuncArgs rf{};
      rf.net = r.net;
      rf.N = KE;
      rf.obs_stride = d.d;
      rf.Kep = d.K;
      rf.pa_stride = (int64_t)d.T * 2;
      rf.tn = c->tn;
      rf.general_t = 0;
      rf.slot = c->slot_dev;
      rf.fre = ws + w.fre;
      rf.fim = ws + w.fim;
      // F travels slot-major between the two kernels of a step: four 128-B runs per store instruction of the MFMA
      // epilogue, one full line per de Hoog load (kernels_ilt.hip, FMODE 2)
      rf.slot_major = 1;
      rf.split = c->md.h == 128 && c->opt_repfunc_split != 0;
      IltArgs ia{nullptr, nullptr, tconst, ws + w.dx, KE, d.d, c->S, c->md.ilt.alpha, std::log(c->md.ilt.tol),
                 c->md.ilt.scale, rf.fre, rf.fim, 1.0, 0, 0, 0, 0, c->eidx_dev};
      StepTailArgs st{};
      st.K = KE;
      st.Kep = d.K;
      st.T = d.T;
      st.nu = d.nu;
      st.d = d.d;
      st.env = d.cost_external ? -1 : d.env;
      st.state_per_sample = state_per_sample;
      st.state0 = state_dev;
      st.x = r.xcarry;
      st.dx = ws + w.dx;
      st.ccarry = r.ccarry;
      st.perturbed = buf->perturbed;
      st.noise = buf->noise;
      st.U = r.U;
      for (int i = 0; i < NLC_MAX_NU * NLC_MAX_NU; ++i) st.sigma_inv[i] = d.noise_sigma_inv[i];
      st.lambda_ = d.lambda_;
      st.u_scale = d.u_scale;
      st.noise_abs_cost = d.noise_abs_cost;
      st.states = buf->states;
      st.cost_total = buf->cost_total;
      // two launches per horizon step: [tail of step t-1 +] representation function -> F, then de Hoog -> dx; the tail
      // of the LAST step is a launch of its own.
      // Round 3: the population is cut into P contiguous parts (multiples of 64 samples) whose 2 T + 1 launches run on P
      // streams.  A sample's chain never leaves its part, so the parts need no ordering among themselves, and the two
      // kernels of a step bound different pipes -- the representation launch the FP64 MFMA (util 0.42: dependent layers),
      // the QD launch the FP64 VALU at 1.25 wavefronts per SIMD (active 0.39) -- so while one part's QD pass runs, another
      // part's representation launch fills the matrix pipe (option "dehoog_streams": 1 = one stream, as before).
      int P = d.E == 1 ? c->opt_dehoog_streams : 1;
      if (P == 0) P = KE >= 8192 ? 2 : 1;  // auto
      if (P > 4) P = 4;
      while (P > 1 && KE / P < 1024) --P;
      if (P > 1) {
        while ((int)c->aux_streams.size() < P - 1) {
          hipStream_t s2 = nullptr;
          NLC_HIP(c, hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
          c->aux_streams.push_back(s2);
        }
        if (!c->ev_fork) NLC_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        while ((int)c->ev_join.size() < P - 1) {
          hipEvent_t e2 = nullptr;
          NLC_HIP(c, hipEventCreateWithFlags(&e2, hipEventDisableTiming));
          c->ev_join.push_back(e2);
        }
        NLC_HIP(c, hipEventRecord(c->ev_fork, c->stream));  // behind the GRU encode and the staged inputs
        for (int h = 1; h < P; ++h) NLC_HIP(c, hipStreamWaitEvent(c->aux_streams[h - 1], c->ev_fork, 0));
      }
      // part h: samples [off_h, off_h + n_h)
      int64_t off_h[4], n_h[4];
      {
        const int64_t per = ((KE / P) + 63) / 64 * 64;
        for (int h = 0; h < P; ++h) {
          off_h[h] = (int64_t)h * per < KE ? (int64_t)h * per : KE;
          n_h[h] = (h == P - 1) ? KE - off_h[h] : (off_h[h] + per <= KE ? per : KE - off_h[h]);
        }
      }
      const size_t f_per_sample = 8 * (size_t)c->net.nt3;
      auto part = [&](int h, RepFuncArgs& rfp, IltArgs& iap, StepTailArgs& stp) {
        const int64_t o = off_h[h], n = n_h[h];
        stp = st;
        stp.K = n;
        stp.Kep = P > 1 ? n : d.K;
        stp.state0 = state_dev + (state_per_sample ? o * d.d : 0);
        stp.x = st.x + o * d.d;
        stp.dx = st.dx + o * d.d;
        stp.ccarry = st.ccarry + o * 2;
        stp.perturbed = st.perturbed + o * d.T * d.nu;
        stp.noise = st.noise + o * d.T * d.nu;
        stp.states = st.states ? st.states + o * d.T * d.d : nullptr;
        stp.cost_total = st.cost_total + o;
        rfp = rf;
        rfp.N = n;
        rfp.Kep = stp.Kep;
        rfp.fre = rf.fre + (size_t)o * f_per_sample;  // slot-major (8 nt3, n) block of the part
        rfp.fim = rf.fim + (size_t)o * f_per_sample;
        iap = ia;
        iap.N = n;
        iap.x = ia.x + o * d.d;
        iap.fre = rfp.fre;
        iap.fim = rfp.fim;
      };
      RepFuncArgs rfs[4];
      IltArgs ias[4];
      StepTailArgs sts[4];
      for (int h = 0; h < P; ++h) part(h, rfs[h], ias[h], sts[h]);
      for (int t = 0; t < d.T; ++t) {
        for (int h = 0; h < P; ++h) {
          if (n_h[h] <= 0) continue;
          hipStream_t sh = h == 0 ? c->stream : c->aux_streams[h - 1];
          if (C > 1 && t % Tc == 0) NLC_HIP(c, hipStreamWaitEvent(sh, c->ev_gru[t / Tc], 0));  // latents of this chunk
          RepFuncArgs& rp = rfs[h];
          rp.obs = sts[h].state0;
          rp.obs_per_sample = (t == 0) ? state_per_sample : 1;
          rp.pa = pa + ((size_t)off_h[h] * d.T + (size_t)t) * 2;
          rp.tail_prev = t > 0;
          if (t > 0) {
            rp.tail = sts[h];
            rp.tail.t = t - 1;
            rp.tail.first = t - 1 == 0;
            rp.tail.last = 0;
          }
          {
            ProfScope ps(c, "nl_repfunc_kernel", sh, true);
            NLC_HIP(c, launch_nl_repfunc(rp, sh));
          }
          if (dehoog) {
            ProfScope ps(c, "ilt_dehoog_kernel", sh, true);
            NLC_HIP(c, launch_ilt_dehoog(ias[h], sh));
          } else {
            const IltArgs& ih = ias[h];
            const IltLinSlotArgs il{ih.fre, ih.fim, ih.eidx, ih.t, ih.x, ih.N, ih.d, ih.S, lin_tab + 2 * ih.S, lin_tab + 3 * ih.S};
            ProfScope ps(c, "ilt_linear_slot_kernel", sh, true);
            NLC_HIP(c, launch_ilt_linear_slot(il, sh));
          }
        }
      }
      for (int h = 0; h < P; ++h) {
        if (n_h[h] <= 0) continue;
        hipStream_t sh = h == 0 ? c->stream : c->aux_streams[h - 1];
        sts[h].t = d.T - 1;
        sts[h].first = d.T == 1;
        sts[h].last = 1;
        {
          ProfScope ps(c, "step_tail_kernel", sh, true);
          NLC_HIP(c, launch_step_tail(sts[h], sh));
        }
        if (h > 0) {
          NLC_HIP(c, hipEventRecord(c->ev_join[h - 1], sh));
          NLC_HIP(c, hipStreamWaitEvent(c->stream, c->ev_join[h - 1], 0));
        }
      }
      return d.cost_external ? NLC_OK : run_weights(c, buf);
    }
    // rollout_variant (nlc_set_option): 0 auto, 1 wave-per-tile, 2 latency-split, 3 fused one-launch body
    int variant = c->opt_rollout_variant;
    const int h_ = c->md.h;
    const bool fused_ok = (h_ == 64 || h_ == 128 || h_ == 256) && 2 * c->g == h_ && c->net.nt3 <= 21 &&
                          KE * d.T * 16 < (int64_t)1 << 31 && !lin_direct;  // (no LIN instance of the one-launch body)
    if (variant == 3 && !fused_ok) return fail(c, NLC_ERR_UNSUPPORTED, "fused planner body: model shape not instantiated");
    // instances per width: 3 and 4 workgroups per CU at hidden_units 64 / 128, 2 at 256 (68 KB of LDS per workgroup)
    const int bpc_hi = h_ == 256 ? 2 : 4, bpc_lo = h_ == 256 ? 2 : 3;
    if (fused_ok && (c->fused_blocks_per_cu < 0 || c->fused_occ_h != h_)) {
      int bpc = 0;
      NLC_HIP(c, fused_max_resident_blocks(h_, bpc_hi, &bpc));
      c->fused_blocks_per_cu = bpc;
      NLC_HIP(c, fused_max_resident_blocks(h_, bpc_lo, &bpc));
      c->fused_blocks_per_cu3 = bpc;
      c->fused_occ_h = h_;
    }
    // The fused body's rollout workgroups wait for encoder workgroups of the SAME launch, so every workgroup must be
    // resident and there must be workgroups left to encode beside one chain per CU: at least two per CU.  It
    // also assumes the device to itself (include/nlc.h): after one hand-off timeout the ctx stays on the two-launch body.
    if (variant == 3 && c->fused_blocks_per_cu < 2)
      return fail(c, NLC_ERR_UNSUPPORTED, "fused planner body: fewer than two workgroups of the kernel fit a CU");
    if (variant == 0 && fused_ok && c->fused_blocks_per_cu >= 2 && KE <= c->opt_fused_max_samples) variant = 3;
    if (variant == 3 && (replay || c->fused_lost)) variant = 2;
    if (variant == 3) {
      const int ncu = c->prop.multiProcessorCount;
      const int ntk_all = (int)((KE + 15) / 16);
      // instance: three workgroups per CU (168 VGPRs) while chains sit on at most half of the CUs, else four (128 VGPRs)
      int built = c->opt_fused_blocks_per_cu ? c->opt_fused_blocks_per_cu : (2 * ntk_all <= ncu ? 3 : 4);
      if (built == 3 && c->fused_blocks_per_cu3 < 3) built = 4;
      if (h_ == 256) built = 2;
      const int bpc = built == bpc_lo ? (c->fused_blocks_per_cu3 < bpc_lo ? c->fused_blocks_per_cu3 : bpc_lo)
                                      : (c->fused_blocks_per_cu < bpc_hi ? c->fused_blocks_per_cu : bpc_hi);
      FusedArgs f{};
      f.r = r;
      f.r.t_begin = 0;
      f.r.t_end = d.T;
      f.g = g;
      f.g.t0 = 0;
      f.g.Tc = d.T;
      f.g.N = KE * d.T;
      FusedCtl& fc = f.ctl;
      fc.sync = reinterpret_cast<unsigned*>(ws + w.sync);
      fc.timeout_host = reinterpret_cast<unsigned*>(fused_timeout_word(c));
      fc.ntk = (int)((KE + 15) / 16);
      fc.n_enc = fc.ntk * d.T;
      // rollout workgroups start one per CU on the first CUs to arrive; by default on half the CUs at most
      // chains start on distinct CUs, one per 16-sample tile (the tiles beyond the CU count drain after the encoders)
      fc.roll_cap = c->opt_fused_roll_cap > 0 ? c->opt_fused_roll_cap : ncu;
      if (fc.roll_cap > fc.ntk) fc.roll_cap = fc.ntk;
      // Schedule (profiles/r2_fused_small_shard.md).  Every workgroup -- the chains' too -- encodes one tile first.  A
      // chain's CU partners then encode M - 1 more tiles each and sleep until the chain is done: with few chains the CUs
      // WITHOUT one feed them alone (M = 1); the more CUs walk a chain, the longer their partners have to help.  M is an
      // empirical fit to the best schedule measured on the MI355X at T = 40 (chains on 25 / 37.5 / 43.75 / 50 % of the
      // CUs, K = 1024 / 1536 / 1792 / 2048: M = 1 / 2 / 3 / 4; e.g. 0.672 ms at K = 2048 against 0.723 without any of
      // this and 0.846 with M = 1), scaled with the horizon.
      const double f_chain = (double)fc.roll_cap / (double)ncu;
      const double extra = (16.0 * f_chain - 4.5) * (double)d.T / 40.0;
      int auto_partner = 1 + (extra > 0 ? (int)extra : 0);
      if (built <= 3) {
        // two partners per chain CU instead of three: measured best M = 1 / 1 / 2 / 6 at chains on 12.5 / 25 / 37.5 / 50 %
        // of the CUs (K = 512 / 1024 / 1536 / 2048, T = 40; 0.521 / 0.527 / 0.563 / 0.674 ms per launch)
        // (K = 1280 / 1792, 31 / 44 %: M = 1 / 4; linear in between)
        const double m3 = f_chain <= 0.3125 ? 1.0 : 1.0 + 26.7 * (f_chain - 0.3125);
        auto_partner = (int)(1.0 + (m3 - 1.0) * (double)d.T / 40.0);
      }
      // experiment (round 3, option "fused_tile_step_ratio" > 0): partners sleep unless the chain-free CUs alone would
      // finish the remaining encoder tiles later than the chain finishes its remaining steps (the kernel's feedback
      // rule).  Measured SLOWER than the static schedule at every K (K = 2048: 0.83 vs 0.67 ms): the rule balances the
      // finishing times but not the ORDER -- the chains consume a horizon step per 11.5 us, the chain-free CUs produce one
      // per 15 us, so the chains starve behind the encoder front while their partners sleep; default off.
      const bool adaptive = c->opt_fused_partner_tiles == -2 && c->opt_fused_tile_step_ratio > 0.0;
      fc.adaptive_q8 = adaptive ? (int)(256.0 * c->opt_fused_tile_step_ratio) : 0;
      fc.pool_wgs = (ncu - fc.roll_cap) * bpc;
      if (adaptive) auto_partner = 1;  // every partner encodes one tile first, then the rule decides
      fc.chain_first_tiles = c->opt_fused_chain_first_tiles >= 0 ? c->opt_fused_chain_first_tiles : 1;
      const int partner = c->opt_fused_partner_tiles >= -1 ? c->opt_fused_partner_tiles : auto_partner;
      // (sleepers need CUs without a chain to produce the latents the chains wait for)
      fc.partner_tiles = (adaptive || fc.roll_cap <= ncu / 2) ? partner : -1;
      fc.spin_limit = (unsigned)c->opt_fused_spin_limit;
      fc.test_drop_tile = c->opt_fused_test_drop_tile;
      // Single planner: the weight reduction runs inside the launch, and with device noise and the command's inputs in
      // the kernel arguments so does the sampling -- command() is then this launch + merge_kernel.
      fc.inline_weights = (c->opt_fused_inline & 1) && d.E == 1 && !d.cost_external;
      fc.inline_perturb = (c->opt_fused_inline & 2) && d.E == 1 && inline_inputs && rng == 1 && d.B <= kFusedMaxInlineB;
      unsigned* sync_words = reinterpret_cast<unsigned*>(ws + w.sync);
      const size_t n_sync = fused_sync_words(d.T, KE);
      if (fc.inline_perturb) {
        // tickets / flags start at zero: the previous command's merge kernel left them so (else: one memset)
        if (c->sync_clean_ws != buf->workspace) NLC_HIP(c, hipMemsetAsync(sync_words, 0, n_sync * sizeof(unsigned), c->stream));
      } else {
        p.zero_words = sync_words;
        p.n_zero_words = (int64_t)n_sync;
        if (int rc = launch_shift_perturb()) return rc;
      }
      c->sync_clean_ws = nullptr;
      c->sync_dirty = true;
      c->last.fused = true;
      f.p = p;
      if (fc.inline_weights) f.w = make_weight_args(c, buf);
      // every workgroup must be resident at once: a rollout workgroup waits for encoder workgroups of the same launch
      const unsigned grid = (unsigned)(ncu * bpc);
      {
        ProfScope ps(c, "nl_plan_fused_kernel");
        NLC_HIP(c, launch_nl_plan_fused(f, c->g, grid, built, c->stream));
      }
      if (fc.inline_weights) return NLC_OK;
    } else {
      if (int rc = launch_shift_perturb()) return rc;
      // Horizon chunks (round 3, option "horizon_chunks"): the encoder input does not depend on the state, so the GRU
      // encode of horizon steps [c Tc, (c + 1) Tc) can run on a stream of its own while the rollout walks the chunk before
      // it.  The two kernels bound different things -- the encoder the FP64 MFMA pipe (95 % busy, two waves per SIMD), the
      // wave-per-tile rollout the latency of ONE wave per SIMD (85 % busy) -- and since round 3 one wave of each fits a
      // SIMD's registers (216 + 288 <= 512), so the encoder fills the rollout's issue bubbles.  State and cost sums travel
      // between the rollout's chunk launches in xcarry / ccarry: same bits as the single launch.
      int C = (variant != 2 && KE > 8192 && d.E == 1) ? c->opt_horizon_chunks : 1;
      if (C < 1) C = 1;
      if (C > d.T) C = d.T;
      if (C > 8) C = 8;
      if (C == 1) {
        g.t0 = 0;
        g.Tc = d.T;
        g.N = KE * d.T;
        {
          ProfScope ps(c, "gru_encode_kernel");
          NLC_HIP(c, launch_gru_encode(g, c->g, c->stream, gru_use_coop(c, g.N)));
        }
        r.t_begin = 0;
        r.t_end = d.T;
        ProfScope ps(c, "nl_rollout_kernel");
        NLC_HIP(c, launch_nl_rollout(r, c->stream, variant));
      } else {
        const int Tc = (d.T + C - 1) / C;
        if (!c->gru_stream) {
          int lo = 0, hi = 0;
          (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = lowest priority: the rollout's workgroups go first
          NLC_HIP(c, hipStreamCreateWithPriority(&c->gru_stream, hipStreamNonBlocking, lo));
        }
        if (!c->ev_fork) NLC_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        while ((int)c->ev_gru.size() < C) {
          hipEvent_t e2 = nullptr;
          NLC_HIP(c, hipEventCreateWithFlags(&e2, hipEventDisableTiming));
          c->ev_gru.push_back(e2);
        }
        NLC_HIP(c, hipEventRecord(c->ev_fork, c->stream));  // behind the perturb kernel and the staged inputs
        NLC_HIP(c, hipStreamWaitEvent(c->gru_stream, c->ev_fork, 0));
        for (int ch = 0; ch < C; ++ch) {
          g.t0 = ch * Tc;
          g.Tc = (g.t0 + Tc <= d.T) ? Tc : d.T - g.t0;
          if (g.Tc <= 0) break;
          g.N = KE * g.Tc;
          {
            ProfScope ps(c, "gru_encode_kernel", c->gru_stream, true);
            NLC_HIP(c, launch_gru_encode(g, c->g, c->gru_stream, false));
          }
          NLC_HIP(c, hipEventRecord(c->ev_gru[ch], c->gru_stream));
        }
        for (int ch = 0; ch < C; ++ch) {
          r.t_begin = ch * Tc;
          r.t_end = (r.t_begin + Tc <= d.T) ? r.t_begin + Tc : d.T;
          if (r.t_begin >= r.t_end) break;
          NLC_HIP(c, hipStreamWaitEvent(c->stream, c->ev_gru[ch], 0));
          ProfScope ps(c, "nl_rollout_kernel");
          NLC_HIP(c, launch_nl_rollout(r, c->stream, 1));
        }
      }
    }
  } else if (d.dynamics == NLC_DYN_NODE) {
    NodeRolloutArgs r{};
    r.net = c->node;
    r.net.nsub = node_substeps(d.ts_pred / c->nd.time_div, c->nd.step_size, r.net.hsub, 8);
    r.K = KE;
    r.Kep = d.K;
    r.T = d.T;
    r.nu = d.nu;
    r.env = d.cost_external ? -1 : d.env;
    r.state_per_sample = state_per_sample;
    r.state0 = state_dev;
    r.perturbed = buf->perturbed;
    r.noise = buf->noise;
    r.U = c->U[c->ucur];
    for (int i = 0; i < NLC_MAX_NU * NLC_MAX_NU; ++i) r.sigma_inv[i] = d.noise_sigma_inv[i];
    r.lambda_ = d.lambda_;
    r.u_scale = d.u_scale;
    r.noise_abs_cost = d.noise_abs_cost;
    r.states = buf->states;
    r.cost_total = buf->cost_total;
    ProfScope ps(c, "node_rollout_kernel");
    NLC_HIP(c, launch_node_rollout(r, c->node_ht, c->stream));
  } else if (d.dynamics == NLC_DYN_DTRNN) {
    RnnArgs g = c->rnn;
    g.mode = 1;
    g.perturbed = buf->perturbed;
    g.abuf = abuf_dev;
    g.u_scale = d.u_scale;
    g.T = d.T;
    g.B = d.B;
    g.Kep = d.K;
    g.K = KE;
    g.N = KE * d.T;
    g.out = ws + w.rq;
    {
      ProfScope ps(c, "rnn_encode_kernel");
      NLC_HIP(c, launch_rnn_encode(g, c->rd.hidden, c->stream));
    }
    RnnRolloutArgs r{};
    r.head = c->rnn_head;
    r.K = KE;
    r.Kep = d.K;
    r.T = d.T;
    r.nu = d.nu;
    r.env = d.cost_external ? -1 : d.env;
    r.state_per_sample = state_per_sample;
    r.state0 = state_dev;
    r.q = ws + w.rq;
    r.perturbed = buf->perturbed;
    r.noise = buf->noise;
    r.U = c->U[c->ucur];
    for (int i = 0; i < NLC_MAX_NU * NLC_MAX_NU; ++i) r.sigma_inv[i] = d.noise_sigma_inv[i];
    r.lambda_ = d.lambda_;
    r.u_scale = d.u_scale;
    r.ts = d.ts_pred;
    r.noise_abs_cost = d.noise_abs_cost;
    r.states = buf->states;
    r.cost_total = buf->cost_total;
    ProfScope ps(c, "rnn_rollout_kernel");
    NLC_HIP(c, launch_rnn_rollout(r, c->stream));
  } else {
    OracleRolloutArgs r{};
    r.K = KE;
    r.Kep = d.K;
    r.T = d.T;
    r.nu = d.nu;
    r.B = d.B;
    r.d = d.d;
    r.env = d.env;
    r.cost_env = d.cost_external ? -1 : d.env;
    r.delay = d.delay;
    r.friction = d.friction;
    r.state_per_sample = state_per_sample;
    r.state0 = state_dev;
    r.abuf = abuf_dev;
    r.perturbed = buf->perturbed;
    r.noise = buf->noise;
    r.U = c->U[c->ucur];
    for (int i = 0; i < NLC_MAX_NU * NLC_MAX_NU; ++i) r.sigma_inv[i] = d.noise_sigma_inv[i];
    r.lambda_ = d.lambda_;
    r.u_scale = d.u_scale;
    r.ts = d.ts_pred;
    r.noise_abs_cost = d.noise_abs_cost;
    r.states = buf->states;
    r.cost_total = buf->cost_total;
    ProfScope ps(c, "oracle_rollout_kernel");
    NLC_HIP(c, launch_oracle_rollout(r, c->stream));
  }
  return d.cost_external ? NLC_OK : run_weights(c, buf);
  NLC_GUARD_END(c)
}

extern "C" int nlc_mppi_rollout(nlc_ctx* c, const double* state, int state_per_sample, const double* abuf_host,
                                const nlc_mppi_buffers* buf, int rng, uint64_t seed, uint64_t counter) {
  if (!c) return NLC_ERR_BAD_ARG;
  return mppi_rollout_impl(c, state, state_per_sample, abuf_host, buf, rng, seed, counter, false);
}

extern "C" int nlc_mppi_weights(nlc_ctx* c, const nlc_mppi_buffers* buf) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->has_mppi) return fail(c, NLC_ERR_STATE, "planner not configured");
  if (!buf || !buf->noise || !buf->cost_total || !buf->cost_nz || !buf->partials || !buf->workspace)
    return fail(c, NLC_ERR_BAD_ARG, "NULL required device buffer");
  NLC_HIP(c, hipSetDevice(c->device));
  return run_weights(c, buf);
  NLC_GUARD_END(c)
}

extern "C" int nlc_mppi_finish(nlc_ctx* c, const double* gathered, int G, int rank, const nlc_mppi_buffers* buf,
                               double* action_host) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!c->has_mppi) return fail(c, NLC_ERR_STATE, "planner not configured");
  if (!buf || !buf->cost_nz) return fail(c, NLC_ERR_BAD_ARG, "NULL argument");
  if (!action_host && !buf->action) return fail(c, NLC_ERR_BAD_ARG, "neither action_host nor buf->action given");
  if (G < 1 || rank < 0 || rank >= G) return fail(c, NLC_ERR_BAD_ARG, "bad G / rank");
  const nlc_mppi_desc& d = c->pd;
  NLC_HIP(c, hipSetDevice(c->device));
  if (!gathered) {
    // the library's own collective: one all-gather of this rank's partials on the command's stream
    if (!c->comm) return fail(c, NLC_ERR_BAD_ARG, "gathered_dev is NULL and no communicator (nlc_comm_init)");
    if (G != c->comm_world || rank != c->comm_rank)
      return fail(c, NLC_ERR_BAD_ARG, "G / rank differ from the communicator's");
    if (!buf->partials) return fail(c, NLC_ERR_BAD_ARG, "NULL buf->partials");
    const size_t per_rank = (size_t)d.E * (size_t)(2 + d.T * d.nu);
    if (c->comm_gather_n < per_rank * (size_t)G) {
      if (c->comm_gather) NLC_HIP(c, hipFree(c->comm_gather));
      c->comm_gather = nullptr;
      NLC_HIP(c, hipMalloc((void**)&c->comm_gather, per_rank * (size_t)G * sizeof(double)));
      c->comm_gather_n = per_rank * (size_t)G;
    }
    ProfScope ps(c, "rccl_all_gather");
    const int rc = rccl()->AllGather(buf->partials, c->comm_gather, per_rank, kNcclFloat64, c->comm, c->stream);
    if (rc != 0) return fail(c, NLC_ERR_COMM, std::string("ncclAllGather: ") + rccl()->GetErrorString(rc));
    gathered = c->comm_gather;
  }
  MergeArgs m{};
  m.Kep = d.K;
  m.E = d.E;
  m.T = d.T;
  m.nu = d.nu;
  m.G = G;
  m.rank = rank;
  m.u_per_command = d.u_per_command;
  m.lambda_ = d.lambda_;
  m.u_scale = d.u_scale;
  m.gathered = gathered;
  m.U = c->U[c->ucur];
  m.cost_nz = buf->cost_nz;
  m.omega = buf->omega;
  m.action = buf->action ? buf->action : c->small;
  m.beta_eta = c->small + (size_t)d.E * d.T * d.nu;
  if (!buf->cost_total) return fail(c, NLC_ERR_BAD_ARG, "NULL buf->cost_total");
  m.cost = buf->cost_total;
  // the returned action is stored by the kernel straight into pinned (host-coherent) memory: the only thing left
  // on the host's critical path is the stream synchronisation
  double* pin_act = c->pinned + (size_t)d.E * d.d + (size_t)d.E * d.B * d.nu;
  m.action_pinned = action_host ? pin_act : nullptr;
  // host wait: spin on a pinned sequence word (single planner) or hipStreamSynchronize
  unsigned long long* seq_word = reinterpret_cast<unsigned long long*>(fused_timeout_word(c) + 1);
  const bool spin = action_host && c->opt_host_spin && d.E == 1 && !c->profiling;
  auto wait_for_action = [&]() -> int {
    if (spin) {
      const unsigned long long want = c->host_seq;
      const volatile unsigned long long* w = seq_word;
      for (unsigned long long it = 0; it < 400000000ull; ++it) {  // ~ seconds: then fall back to the runtime's wait
        if (*w == want) return NLC_OK;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
    }
    NLC_HIP(c, hipStreamSynchronize(c->stream));
    return NLC_OK;
  };
  auto launch_merge_now = [&]() -> int {
    if (spin) {
      m.seq_pinned = seq_word;
      m.seq = ++c->host_seq;
    }
    if (c->sync_dirty && buf->workspace && !c->opt_fused_keep_sync) {
      // last launch of the command: leave the fused body's tickets / flags zeroed for the next one
      m.zero_words = reinterpret_cast<unsigned*>((double*)buf->workspace + ws_layout(c).sync);
      m.n_zero_words = (int64_t)fused_sync_words(d.T, d.K * d.E);
    }
    ProfScope ps(c, "merge_kernel");
    NLC_HIP(c, launch_merge(m, c->stream));
    if (m.zero_words) {
      c->sync_clean_ws = buf->workspace;
      c->sync_dirty = false;
    }
    return NLC_OK;
  };
  if (int rc = launch_merge_now()) return rc;
  if (action_host) {
    const size_t na = (size_t)d.E * d.u_per_command * d.nu;
    if (int rc = wait_for_action()) return rc;
    if (fused_gave_up(c)) {
      // A wave of the fused body gave up waiting for another workgroup of its launch (the device was not this planner's
      // alone, or fewer workgroups were resident than the host assumed): the command's result is not valid.  This ctx
      // runs the two-launch body from now on; a single-rank command with library-side costs is re-run on it right here.
      c->fused_lost = true;
      if (G != 1 || d.cost_external || !c->last.valid || !c->last.fused || !buf->workspace || !buf->partials)
        return fail(c, NLC_ERR_HIP, "fused planner body: a workgroup timed out waiting inside the launch; command lost "
                                    "(later commands run the two-launch body)");
      c->fused_fallbacks += 1;
      c->ucur ^= 1;  // back to the control sequence before this command's shift (the ping-pong partner is untouched)
      const nlc_ctx::LastCommand L = c->last;
      if (int rc = mppi_rollout_impl(c, L.state_in, L.state_per_sample, L.abuf_in, buf, L.rng, L.seed, L.counter, true)) return rc;
      m.U = c->U[c->ucur];
      if (gathered == nullptr || G == 1) m.gathered = buf->partials;
      if (int rc = launch_merge_now()) return rc;
      if (int rc = wait_for_action()) return rc;
    }
    std::memcpy(action_host, pin_act, na * sizeof(double));
  }
  return NLC_OK;
  NLC_GUARD_END(c)
}

// =================================================================================== profiling
extern "C" int nlc_profile_enable(nlc_ctx* c, int on) {
  if (!c) return NLC_ERR_BAD_ARG;
  if (!on) prof_flush(c);
  c->profiling = on != 0;
  return NLC_OK;
}
extern "C" int nlc_profile_reset(nlc_ctx* c) {
  if (!c) return NLC_ERR_BAD_ARG;
  prof_flush(c);
  c->prof.clear();
  return NLC_OK;
}
extern "C" int nlc_profile_count(nlc_ctx* c) {
  if (!c) return NLC_ERR_BAD_ARG;
  return (int)c->prof.size();
}
extern "C" int nlc_profile_read(nlc_ctx* c, int idx, char* name, int name_len, double* total_ms, int64_t* launches) {
  if (!c) return NLC_ERR_BAD_ARG;
  if (idx < 0 || idx >= (int)c->prof.size()) return fail(c, NLC_ERR_BAD_ARG, "profile index out of range");
  prof_flush(c);
  const ProfEntry& p = c->prof[idx];
  if (name && name_len > 0) std::snprintf(name, name_len, "%s", p.name.c_str());
  if (total_ms) *total_ms = p.total_ms;
  if (launches) *launches = p.launches;
  return NLC_OK;
}
