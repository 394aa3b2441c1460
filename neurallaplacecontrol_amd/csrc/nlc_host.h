// Internal header of the host side of libnlc_hip.so (the abi_*.hip translation units): the context object behind the opaque
// nlc_ctx handle of include/nlc.h, status / HIP-error helpers, per-launch event profiling, and the few helpers more than one
// unit needs.  Nothing here is part of the ABI.
//   abi_ctx.hip         context, options, stream binding, device info, profiling read-out
//   abi_comm.hip        optional library-owned RCCL communicator (bound with dlopen)
//   abi_ilt.hip         stand-alone ILT entry points (laplace_reconstruct pieces) + tables of the linear algorithms
//   abi_model.hip       NL model upload (MFMA fragment packing), GRU encode, model forward, representation function
//   abi_baselines.hip   env step, Delta-t RNN and NODE baseline models
//   abi_planner.hip     nlc_mppi_*: configure, workspace layout, phase 1 (sampling + rollout), weights, finish
//   abi_planner_nl.hip  phase 1 with Neural-Laplace dynamics: staged (de Hoog / linear), one-launch fused body, two launches
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <time.h>

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

namespace host {

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

}  // namespace host
}  // namespace nlc

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
  nlc::host::DeviceArena arena;
  nlc::GruArgs gru{};   // weight pointers + normalisation filled in
  nlc::NlNetArgs net{}; // general-t variant (b1 = raw bias)
  std::vector<double> W1s_host, b1_host;  // for folding the constant sphere inputs at configure time
  int* slot_dev = nullptr;                // (8*nt3) layer-3 slot -> c*S + k (de Hoog path)
  double* lin_tab = nullptr;              // nodes / weights of the linear ILT algorithms, cached per (algo, terms)
  int lin_algo = -1, lin_S = 0;
  int* eidx_dev = nullptr;                // (d*S) inverse: term k of dim c -> slot (slot-major F of the planner path)

  // Delta-t RNN baseline model
  bool has_rnn = false;
  nlc_rnn_desc rd{};
  double* rnn_base = nullptr;  // packed weights (device)
  nlc::RnnArgs rnn{};
  nlc::RnnHead rnn_head{};

  // NODE baseline model
  bool has_node = false;
  nlc_node_desc nd{};
  double* node_base = nullptr;
  nlc::NodeNetArgs node{};
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
  std::vector<nlc::host::ProfEntry> prof;
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
  int64_t fused_timeouts = 0;           // fused launches of THIS ctx that gave up (re-run or lost)
  int64_t commands = 0;                 // nlc_mppi_rollout calls since nlc_create
  int64_t last_giveup_command = -1;     // value of `commands` (0-based) at the last give-up seen by this ctx, -1 = none
  int last_body = 0;                    // body phase 1 of the last command ran on (nlc_get_stat "rollout_body")
  // tools only (tools/rollout_giveback.py): something BETWEEN the encoder launch and the rollout launch of the two-launch body
  double opt_dbg_gap_us = 0.0;          // one wavefront spinning on the constant 100 MHz counter for this long (an idle GPU)
  double opt_dbg_l2_mb = 0.0;           // a read sweep over this many MB of scratch (evicts the L2s)
  double* dbg_scratch = nullptr;
  size_t dbg_scratch_bytes = 0;
  int opt_gru_gemm = 0;                 // 1: encoder hidden-state GEMMs on the INT8 matrix pipe (kernels_gru_i8.hip), g == 64 only
  int opt_horizon_chunks = 1;           // Fourier planner, wave-per-tile body (K > 8192): GRU encode of later horizon chunks beside the rollout of earlier ones
  int opt_dehoog_gru_chunks = 0;        // staged de Hoog planner: GRU encode in this many horizon chunks beside the step chain (0 / 1: one launch up front)
  int opt_dehoog_gru_lds_pad = 49152;   // unused dynamic LDS of those chunk launches (bytes): 32 KB + 48 KB -> two workgroups per CU
  hipStream_t gru_stream = nullptr;
  std::vector<hipEvent_t> ev_gru;
  int opt_dehoog_chain = -1;            // de Hoog planner: the step chain as one persistent launch (kernels_dehoog_chain.hip): -1 auto, 0 / 1
  // de Hoog planner, both knobs on auto: the chain's form is measured over the planner's first commands (abi_planner_nl.hip)
  int dh_n = 0, dh_choice = -1, dh_pending = -1, dh_pending_round = 0;
  double dh_t0 = 0.0;
  float dh_ms[3][2] = {{1e30f, 1e30f}, {1e30f, 1e30f}, {1e30f, 1e30f}};  // [candidate][round & 1]: the last two rounds
  hipEvent_t dh_ev[2] = {nullptr, nullptr};
  int opt_dehoog_chain_phases = 3;      // tools only: 1 / 2 = only the representation / QD phase of the chain kernel runs (timing)
  int opt_dehoog_streams = 0;           // staged de Hoog planner: parts of the population on streams of their own (0 auto)
  std::vector<hipStream_t> aux_streams;
  hipEvent_t ev_fork = nullptr;
  std::vector<hipEvent_t> ev_join;
  double opt_fused_tile_step_ratio = 0.0;  // > 0: the adaptive partner rule (measured slower: profiles/r3_fused_small_shard.md); 0 = static schedule
  int opt_host_spin = 1;                // nlc_mppi_finish with a host action pointer: 1 spin on a pinned word the merge kernel
                                        // stores, 2 sleep through the predicted wait first, 0 hipStreamSynchronize
  unsigned long long host_seq = 0;      // sequence number of the last command handed to the spin protocol
  double opt_host_spin_margin_us = 150.0;  // host_spin 2: the host wakes this long (or 15 % of the predicted wait) before the predicted end
  double wait_hist_us[8] = {0};         // host_spin: the last waits for the action (their minimum predicts the next one)
  int wait_hist_n = 0, wait_hist_at = 0;
  double nap_margin_us = 0.0;           // host_spin 2: the margin in use (grows when a nap overshoots the action's arrival)
  int opt_fused_inline = 3;             // fused body: sampling / bounding and the weight reduction inside the launch
  int64_t opt_fused_spin_limit = 1 << 18;  // polls (~2 us each) before a waiting wave of the fused body gives up (~0.5 s)
  int opt_fused_test_drop_tile = -1;    // tests only: this encoder tile is never published (forces the timeout path)
  double opt_test_lin_coeff_scale = 1.0;  // tests only: the largest w_re / t coefficient of the LIN fragments is scaled by this
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
    double abuf_in[nlc::kMaxInlineAbuf] = {0};
  } last;
  // optional native collective (nlc_comm_init): an RCCL communicator over the ranks of a K-sharded planner
  void* comm = nullptr;
  int comm_world = 0, comm_rank = 0;
  double* comm_gather = nullptr;  // (world, E, 2+T*nu) receive buffer of the per-command all-gather
  size_t comm_gather_n = 0;
};

namespace nlc {
namespace host {

extern thread_local std::string g_create_error;  // nlc_last_error(NULL): failures before a ctx exists

inline int fail(nlc_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
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

void prof_flush(nlc_ctx* c);

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

bool is_device_ptr(const void* p);
bool gru_use_coop(const nlc_ctx* c, int64_t n_windows);

// ---- abi_ilt.hip
int check_ilt(nlc_ctx* c, const nlc_ilt_desc* d);
void sphere_inputs(const nlc_ilt_desc& ilt, double tn, std::vector<double>& sph);
void linear_tables_host(int algo, int S, std::vector<double>& h);
int linear_tables(nlc_ctx* c, const nlc_ilt_desc* d, const double** tab);

// ---- abi_baselines.hip
int node_substeps(double t_end, double step, double* h, int max_n);

// ---- abi_comm.hip: RCCL, bound at run time
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
Rccl* rccl();
constexpr int kNcclFloat64 = 8;  // rccl.h: ncclFloat64 = ncclDouble = 8

// ---- abi_planner.hip
bool linear_on_rollout_kernels(const nlc_ctx* c);
struct WsLayout {
  size_t tile_part, chunk_part, pa, state0, abuf, xcarry, ccarry, fre, fim, dx, tconst, rq, sync, total;
};
WsLayout ws_layout(const nlc_ctx* c);
double* fused_timeout_word(nlc_ctx* c);
bool fused_gave_up(nlc_ctx* c);
unsigned* merge_status_word(nlc_ctx* c);
bool merge_reported_invalid(nlc_ctx* c);
WeightArgs make_weight_args(nlc_ctx* c, const nlc_mppi_buffers* buf);
int run_weights(nlc_ctx* c, const nlc_mppi_buffers* buf);

// One phase-1 call (nlc_mppi_rollout, or its re-run after a fused time-out) as the dynamics-specific parts see it
struct RolloutCall {
  const nlc_mppi_buffers* buf;
  PerturbArgs p;         // sampling / bounding arguments, filled in by the caller
  WsLayout w;
  double* ws;            // buf->workspace
  double* state_dev;     // staged state(s) and action buffer(s) inside the workspace
  double* abuf_dev;
  int64_t KE;            // all local samples, episode-major
  int state_per_sample, rng;
  bool inline_inputs;    // state and action_buffer ride in the perturb kernel's arguments
  bool replay;           // re-run of the last command on the two-launch body
};
int launch_shift_perturb(nlc_ctx* c, RolloutCall& call);
// ---- abi_planner_nl.hip
int rollout_nl(nlc_ctx* c, RolloutCall& call);

}  // namespace host
}  // namespace nlc
