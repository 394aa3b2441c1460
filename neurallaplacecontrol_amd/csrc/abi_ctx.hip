// Host side of libnlc_hip.so, context unit: nlc_create / nlc_destroy, planner options, stream binding, device info and the
// read-out of the per-launch event profile.  See include/nlc.h for the contract of every entry point.
#include "nlc_host.h"

using namespace nlc;
using namespace nlc::host;

namespace nlc {
namespace host {

thread_local std::string g_create_error;

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

bool is_device_ptr(const void* p) {
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // plain (unregistered) host memory: not an error for us
    return false;
  }
  return at.type == hipMemoryTypeDevice;
}

}  // namespace host
}  // namespace nlc

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

extern "C" void nlc_destroy(nlc_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  nlc_comm_destroy(c);
  prof_flush(c);
  if (c->arena.base) hipFree(c->arena.base);
  if (c->dbg_scratch) hipFree(c->dbg_scratch);
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
  for (hipEvent_t e : c->dh_ev)
    if (e) hipEventDestroy(e);
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
  } else if (n == "dehoog_chain") {
    if (value != -1 && value != 0 && value != 1 && value != 2) return fail(c, NLC_ERR_BAD_ARG, "dehoog_chain must be -1 (auto), 0, 1 or 2");
    c->opt_dehoog_chain = (int)value;
  } else if (n == "dehoog_chain_phases") {
    if (value != 1 && value != 2 && value != 3) return fail(c, NLC_ERR_BAD_ARG, "dehoog_chain_phases must be 1, 2 or 3");
    c->opt_dehoog_chain_phases = (int)value;
  } else if (n == "dehoog_streams") {
    if (value < 0 || value > 4 || value != (int)value) return fail(c, NLC_ERR_BAD_ARG, "dehoog_streams must be 0 (auto), 1, 2, 3 or 4");
    c->opt_dehoog_streams = (int)value;
  } else if (n == "fused_tile_step_ratio") {
    if (value < 0 || value > 64) return fail(c, NLC_ERR_BAD_ARG, "fused_tile_step_ratio must be in 0 .. 64 (0 = static schedule)");
    c->opt_fused_tile_step_ratio = value;
  } else if (n == "host_spin") {
    if (value != 0 && value != 1 && value != 2) return fail(c, NLC_ERR_BAD_ARG, "host_spin must be 0, 1 or 2");
    c->opt_host_spin = (int)value;
    c->wait_hist_n = c->wait_hist_at = 0;
    c->nap_margin_us = 0.0;
  } else if (n == "host_spin_margin_us") {
    if (value < 0 || value > 1.0e6) return fail(c, NLC_ERR_BAD_ARG, "host_spin_margin_us must be in 0 .. 1e6");
    c->opt_host_spin_margin_us = value;
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
  } else if (n == "test_lin_coeff_scale") {
    c->opt_test_lin_coeff_scale = value;
    c->has_mppi = false;  // folded at nlc_mppi_configure
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
  } else if (n == "gru_gemm") {
    // 0: FP64 MFMAs (default); 1: the encoder's hidden-state GEMMs as int8-sliced fixed-point products (kernels_gru_i8.hip;
    // hidden_units = 128: the stand-alone encoder launches that take the wave-sized form; the one-launch planner body keeps its FP64
    // encoder role)
    if (value != 0 && value != 1) return fail(c, NLC_ERR_BAD_ARG, "gru_gemm must be 0 (FP64 MFMA) or 1 (int8-sliced)");
    c->opt_gru_gemm = (int)value;
    if (c->has_model) c->gru.use_i8 = (c->gru.i8_stream != nullptr && value == 1) ? 1 : 0;
  } else if (n == "dbg_gap_us") {
    if (value < 0 || value > 1.0e5) return fail(c, NLC_ERR_BAD_ARG, "dbg_gap_us must be in 0 .. 1e5");
    c->opt_dbg_gap_us = value;
  } else if (n == "dbg_l2_mb") {
    if (value < 0 || value > 4096) return fail(c, NLC_ERR_BAD_ARG, "dbg_l2_mb must be in 0 .. 4096");
    c->opt_dbg_l2_mb = value;
  } else if (n == "fused_max_samples") {
    if (value < 0) return fail(c, NLC_ERR_BAD_ARG, "fused_max_samples must be >= 0");
    c->opt_fused_max_samples = (int64_t)value;
  } else {
    return fail(c, NLC_ERR_BAD_ARG, "unknown option: " + n);
  }
  return NLC_OK;
}

extern "C" int nlc_get_stat(nlc_ctx* c, const char* name, double* out) {
  if (!c) return NLC_ERR_BAD_ARG;
  if (!name || !out) return fail(c, NLC_ERR_BAD_ARG, "NULL stat name / out");
  const std::string n(name);
  if (n == "rollout_body") *out = (double)c->last_body;
  else if (n == "fused_timeouts") *out = (double)c->fused_timeouts;
  else if (n == "fused_fallbacks") *out = (double)c->fused_fallbacks;
  else if (n == "fused_lost") *out = c->fused_lost ? 1.0 : 0.0;
  else if (n == "last_giveup_command") *out = (double)c->last_giveup_command;
  else if (n == "commands") *out = (double)c->commands;
  else if (n == "comm_world") *out = c->comm ? (double)c->comm_world : 0.0;
  else if (n == "comm_rank") *out = c->comm ? (double)c->comm_rank : -1.0;
  else if (n == "fused_blocks_per_cu") *out = (double)c->fused_blocks_per_cu;
  else if (n == "fused_spin_limit") *out = (double)c->opt_fused_spin_limit;
  else if (n == "model_nt3") *out = c->has_model ? (double)c->net.nt3 : 0.0;
  else if (n == "gru_gemm") *out = (c->has_model && c->gru.use_i8) ? 1.0 : 0.0;  // 1: the encoder launches run kernels_gru_i8.hip
  else if (n == "gru_i8_launches") *out = (double)nlc::gru_i8_launch_count();  // process-wide
  else return fail(c, NLC_ERR_BAD_ARG, "unknown stat: " + n);
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
