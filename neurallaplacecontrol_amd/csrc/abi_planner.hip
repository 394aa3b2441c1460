// Host side of libnlc_hip.so, planner unit: MPPIDelay behind nlc_mppi_* (planners/mppi_delay.py:64-224) -- configure,
// workspace layout, phase 1 (sampling / bounding + the T-step rollout; Neural-Laplace dynamics in abi_planner_nl.hip),
// importance weights, and the merge / hand-over of phase 2.
#include "nlc_host.h"

using namespace nlc;
using namespace nlc::host;

extern "C" int nlc_mppi_configure(nlc_ctx* c, const nlc_mppi_desc* d) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (!d) return fail(c, NLC_ERR_BAD_ARG, "NULL desc");
  if (d->K < 1 || d->T < 1 || d->K_global < d->K || d->k_offset < 0 || d->k_offset + d->K > d->K_global)
    return fail(c, NLC_ERR_BAD_SHAPE, "bad K / K_global / k_offset / T");
  if (d->nu < 1 || d->nu > NLC_MAX_NU) return fail(c, NLC_ERR_UNSUPPORTED, "nu must be 1 or 2");
  // (caller-supplied dynamics: the library never touches a state, any nx goes)
  if (d->d < 1 || (d->d > NLC_MAX_D && d->dynamics != NLC_DYN_EXTERNAL)) return fail(c, NLC_ERR_BAD_SHAPE, "bad nx");
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
  if (!env_free && (d->env < 0 || d->env > NLC_ENV_CARTPOLE_NOTRIG)) return fail(c, NLC_ERR_UNSUPPORTED, "unknown env id");
  static const int env_d[4] = {5, 3, 6, 4}, env_nu[4] = {1, 1, 2, 1};
  if (!env_free && d->dynamics != NLC_DYN_EXTERNAL && (d->d != env_d[d->env] || d->nu != env_nu[d->env]))
    return fail(c, NLC_ERR_BAD_SHAPE, "nx / nu do not match the env's observation");
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
      if (c->opt_test_lin_coeff_scale != 1.0) {  // tests only: the sweep's tolerance must catch this
        size_t at = 0;
        for (size_t i = 0; i < cp.size(); ++i)
          if (std::fabs(cp[i]) > std::fabs(cp[at])) at = i;
        cp[at] *= c->opt_test_lin_coeff_scale;  // (the largest weight: Stehfest's span many orders of magnitude)
      }
      if (c->cp_lin) hipFree(c->cp_lin);
      c->cp_lin = nullptr;
      NLC_HIP(c, hipMalloc((void**)&c->cp_lin, cp.size() * sizeof(double)));
      NLC_HIP(c, hipMemcpy(c->cp_lin, cp.data(), cp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  c->has_mppi = true;
  c->dh_n = 0;  // de Hoog chain form: measured again for the new problem
  c->dh_choice = c->dh_pending = -1;
  for (auto& v : c->dh_ms) v[0] = v[1] = 1e30f;
  c->wait_hist_n = c->wait_hist_at = 0;  // host_spin 2: a new problem size, a new wait to predict
  c->nap_margin_us = 0.0;
  c->sync_clean_ws = nullptr;
  c->sync_dirty = false;
  c->last.valid = false;
  return NLC_OK;
  NLC_GUARD_END(c)
}

namespace nlc {
namespace host {
// fixed Talbot / Stehfest models whose rollout runs on the LIN instances of the rollout kernels (kernels_nl_lin*.hip) instead of
// the staged path
bool linear_on_rollout_kernels(const nlc_ctx* c) {
  return c->has_model && (c->md.ilt.algo == NLC_ILT_FIXED_TALBOT || c->md.ilt.algo == NLC_ILT_STEHFEST) &&
         (c->md.h == 64 || c->md.h == 128 || c->md.h == 256) && c->opt_linear_fused != 0;
}

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
  // slot-major (8*nt3, KE), >= KE*d*S; the persistent step chain cuts it into private (8*nt3) x 64 blocks, one per 64 samples
  const size_t KE64 = (KE + 63) / 64 * 64;
  w.fre = take(staged ? KE64 * 8 * (size_t)c->net.nt3 : 0);
  w.fim = take(staged ? KE64 * 8 * (size_t)c->net.nt3 : 0);
  w.dx = take(staged ? KE * d.d : 0);
  w.tconst = take(staged ? 8 : 0);
  w.rq = take(d.dynamics == NLC_DYN_DTRNN ? KE * d.T * d.d : 0);  // hidden part of linear_out, (T, K, d)
  // fused one-launch planner body: tickets, per-CU census, one flag word per encoder tile (unsigned words)
  w.sync = take(d.dynamics == NLC_DYN_NL && !staged ? (fused_sync_words(d.T, (int64_t)KE) + 1) / 2 : 0);
  w.total = off;
  return w;
}

// pinned host word a rollout workgroup of the fused planner body sets when it gives up waiting for an encoder tile
double* fused_timeout_word(nlc_ctx* c) {
  const nlc_mppi_desc& d = c->pd;
  return c->pinned + (size_t)d.E * d.d + (size_t)d.E * d.B * d.nu + (size_t)d.E * d.T * d.nu;
}
bool fused_gave_up(nlc_ctx* c) {
  unsigned* w = reinterpret_cast<unsigned*>(fused_timeout_word(c));
  if (*w == 0u) return false;
  *w = 0u;
  return true;
}
// pinned word merge_kernel sets when a gathered partial row is marked invalid (kPartialInvalidEta): read and cleared here
unsigned* merge_status_word(nlc_ctx* c) { return reinterpret_cast<unsigned*>(fused_timeout_word(c) + 2); }
bool merge_reported_invalid(nlc_ctx* c) {
  unsigned* w = merge_status_word(c);
  if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == 0u) return false;
  *w = 0u;
  return true;
}

WeightArgs make_weight_args(nlc_ctx* c, const nlc_mppi_buffers* buf) {
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

int run_weights(nlc_ctx* c, const nlc_mppi_buffers* buf) {
  WeightArgs wa = make_weight_args(c, buf);
  // behind a fused launch that left the fold to us: its give-up word marks the partial rows here -- the launch of
  // THIS command only; a re-run on the two-launch body (last_body 2) must not read the word the failed launch left set
  if (c->last_body == 3 && buf->workspace && c->pd.dynamics == NLC_DYN_NL)
    wa.gave_up = reinterpret_cast<const unsigned*>((const double*)buf->workspace + ws_layout(c).sync) + kFusedTimeout;
  ProfScope ps(c, "weight_kernels");
  NLC_HIP(c, launch_weights(wa, c->stream));
  return NLC_OK;
}

// one launch: the perturb kernel shifts U on the fly (every (k, t) thread reads U_old[t + 1]; the episode's first local sample
// stores the shifted row) and stores the staged inputs; shift_U_kernel only when there is nothing to perturb
int launch_shift_perturb(nlc_ctx* c, RolloutCall& call) {
  PerturbArgs& p = call.p;
  p.fused_shift = (int64_t)p.K * p.T > 0;
  if (!p.fused_shift) {
    ProfScope ps(c, "shift_U_kernel");
    NLC_HIP(c, launch_shift_U(p, c->stream));
  }
  ProfScope ps(c, "perturb_kernel");
  NLC_HIP(c, launch_perturb(p, c->stream));
  return NLC_OK;
}
}  // namespace host
}  // namespace nlc

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
  if (!replay) c->commands += 1;
  c->last_body = 0;
  if (!replay) {  // (a device-resident caller never synchronised inside nlc_mppi_finish)
    const bool lost_here = fused_gave_up(c), marked = merge_reported_invalid(c);
    if (lost_here) {
      c->fused_lost = true;  // from here on the two-launch body
      c->fused_timeouts += 1;
      c->last_giveup_command = c->commands - 2;  // the command before this one
    }
    if (lost_here || marked)  // `marked` is set on every rank of a sharded planner: they all fail here, none waits in a collective
      return fail(c, NLC_ERR_HIP, "fused planner body: the PREVIOUS command timed out waiting inside the launch (on this or "
                                  "another rank; its action was NaN and U was left alone); later commands run the two-launch body");
  }
  RolloutCall call{};
  call.buf = buf;
  call.w = ws_layout(c);
  call.ws = (double*)buf->workspace;
  call.state_dev = call.ws + call.w.state0;
  call.abuf_dev = call.ws + call.w.abuf;
  call.KE = d.K * d.E;  // all local samples, episode-major
  call.state_per_sample = state_per_sample;
  call.rng = rng;
  call.replay = replay;
  const WsLayout& w = call.w;
  double* ws = call.ws;
  double* state_dev = call.state_dev;
  double* abuf_dev = call.abuf_dev;
  const int64_t KE = call.KE;
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
  PerturbArgs& p = call.p;
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
  c->ucur ^= 1;  // from here on c->U[c->ucur] is the shifted sequence the kernels below produce
  call.inline_inputs = inline_inputs;
  const bool lin_direct = d.dynamics == NLC_DYN_NL && linear_on_rollout_kernels(c);
  const bool nl_fourier = d.dynamics == NLC_DYN_NL && (c->md.ilt.algo == NLC_ILT_FOURIER || lin_direct);  // one rollout kernel
  // (the fused planner body's argument block carries the sampling arguments to the device itself: rollout_nl launches the
  // perturb kernel on the paths that still need it)
  if (!nl_fourier)
    if (int rc2 = launch_shift_perturb(c, call)) return rc2;
  if (external) {
    c->last_body = 9;
    return NLC_OK;  // the caller runs the horizon loop, then nlc_mppi_weights
  }
  if (d.dynamics == NLC_DYN_NL) return rollout_nl(c, call);
  c->last_body = d.dynamics == NLC_DYN_NODE ? 8 : (d.dynamics == NLC_DYN_DTRNN ? 7 : 6);
  if (d.dynamics == NLC_DYN_NODE) {
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

namespace {

// the library's own collective: one all-gather of this rank's partial rows on the command's stream
int native_all_gather(nlc_ctx* c, const nlc_mppi_buffers* buf, int G, int rank, const double** gathered) {
  const nlc_mppi_desc& d = c->pd;
  if (!c->comm) return fail(c, NLC_ERR_BAD_ARG, "gathered_dev is NULL and no communicator (nlc_comm_init)");
  if (G != c->comm_world || rank != c->comm_rank) return fail(c, NLC_ERR_BAD_ARG, "G / rank differ from the communicator's");
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
  *gathered = c->comm_gather;
  return NLC_OK;
}

inline double now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

// Host wait for the action of a single planner.  host_spin 1: spin on the pinned sequence word the merge kernel stores
// behind the action.  host_spin 2: sleep through most of the command first -- the shortest of the last eight waits predicts
// this one -- and spin only for the rest: the core is free for the harness's other workers (run_exp_multi.py:145 fans out a
// Pool(12) over one GPU) at the price of a timer wake-up when the prediction is off.  host_spin 0 / batched planners /
// profiling: hipStreamSynchronize (which, measured, busy-waits as well: profiles/r4_host_spin_contention.json).
int wait_for_action(nlc_ctx* c, bool spin, const unsigned long long* seq_word) {
  if (spin) {
    const unsigned long long want = c->host_seq;
    const double t0 = now_us();
    if (c->opt_host_spin == 2 && c->wait_hist_n >= 2) {
      double pred = c->wait_hist_us[0];
      for (int i = 1; i < c->wait_hist_n; ++i) pred = pred < c->wait_hist_us[i] ? pred : c->wait_hist_us[i];
      double margin = pred * 0.15 > c->opt_host_spin_margin_us ? pred * 0.15 : c->opt_host_spin_margin_us;
      if (c->nap_margin_us > margin) margin = c->nap_margin_us;  // grown by earlier oversleeps (below)
      const double nap = pred - margin;
      if (nap >= 50.0 && __atomic_load_n(seq_word, __ATOMIC_ACQUIRE) != want) {
        timespec req;
        req.tv_sec = (time_t)(nap * 1e-6);
        req.tv_nsec = (long)((nap - (double)req.tv_sec * 1e6) * 1e3);
        nanosleep(&req, nullptr);
        // the timer's wake-up latency is the host's, not ours to know: if the action was already there when we woke, the nap
        // was too long -- wake earlier from now on; while it is not, drift back towards the configured margin
        if (__atomic_load_n(seq_word, __ATOMIC_ACQUIRE) == want)
          c->nap_margin_us = margin * 1.5 + 20.0;
        else
          c->nap_margin_us = margin * 0.995;
      }
    }
    for (unsigned long long it = 0; it < 400000000ull; ++it) {  // ~ seconds: then fall back to the runtime's wait
      // acquire: the action words the kernel stored BEFORE the sequence number are read after it
      if (__atomic_load_n(seq_word, __ATOMIC_ACQUIRE) == want) {
        c->wait_hist_us[c->wait_hist_at] = now_us() - t0;
        c->wait_hist_at = (c->wait_hist_at + 1) % 8;
        if (c->wait_hist_n < 8) c->wait_hist_n += 1;
        return NLC_OK;
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
  }
  NLC_HIP(c, hipStreamSynchronize(c->stream));
  return NLC_OK;
}

}  // namespace

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
  const bool native = gathered == nullptr;
  if (native)
    if (int rc = native_all_gather(c, buf, G, rank, &gathered)) return rc;
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
  // on the host's critical path is the wait for it
  double* pin_act = c->pinned + (size_t)d.E * d.d + (size_t)d.E * d.B * d.nu;
  m.action_pinned = action_host ? pin_act : nullptr;
  m.status_pinned = merge_status_word(c);
  unsigned long long* seq_word = reinterpret_cast<unsigned long long*>(fused_timeout_word(c) + 1);
  const bool spin = action_host && c->opt_host_spin && d.E == 1 && !c->profiling;
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
  if (!action_host) return NLC_OK;  // device-resident caller: no wait; a lost command surfaces at the next nlc_mppi_rollout
  const size_t na = (size_t)d.E * d.u_per_command * d.nu;
  if (int rc = wait_for_action(c, spin, seq_word)) return rc;
  // A wave of the fused body gave up waiting for another workgroup of its launch (the device was not this planner's alone,
  // or fewer workgroups were resident than the host assumed): that launch's results are not valid.
  //   * here: the pinned give-up word is set; this ctx runs the two-launch body from now on;
  //   * on ANY rank of a sharded planner: the shard's partial row is marked (kPartialInvalidEta) and travels through the
  //     all-gather, so merge_kernel on EVERY rank has skipped the update and set the merge-status word.
  // Every rank therefore takes the same decision: re-run the command on the two-launch body from the inputs the ctx keeps
  // (the control sequence's ping-pong partner is untouched), gather again, merge again.
  const bool lost_here = fused_gave_up(c);
  const bool marked = merge_reported_invalid(c);
  if (lost_here) {
    c->fused_lost = true;
    c->fused_timeouts += 1;
  }
  if (lost_here || marked) {
    c->last_giveup_command = c->commands - 1;
    if (G > 1 && !marked)  // (a sharded command whose weights were folded outside the launch: the peers cannot know)
      return fail(c, NLC_ERR_HIP, "fused planner body: a workgroup timed out waiting inside the launch and the shard's partials "
                                  "were not marked; command lost (later commands run the two-launch body)");
    if (d.cost_external || !c->last.valid || !buf->workspace || !buf->partials)
      return fail(c, NLC_ERR_HIP, "fused planner body: a workgroup timed out waiting inside the launch; command lost "
                                  "(later commands run the two-launch body)");
    c->fused_fallbacks += 1;
    c->ucur ^= 1;  // back to the control sequence before this command's shift
    const nlc_ctx::LastCommand L = c->last;
    if (int rc = mppi_rollout_impl(c, L.state_in, L.state_per_sample, L.abuf_in, buf, L.rng, L.seed, L.counter, true)) return rc;
    m.U = c->U[c->ucur];
    if (G == 1) {
      m.gathered = buf->partials;
    } else if (native) {
      if (int rc = native_all_gather(c, buf, G, rank, &gathered)) return rc;
      m.gathered = gathered;
    } else {
      // the caller owns the collective: buf->partials hold the re-run's values -- gather them again, call again
      c->err = "the command was re-run on the two-launch body: all-gather buf->partials again and call nlc_mppi_finish again";
      return NLC_AGAIN;
    }
    if (int rc = launch_merge_now()) return rc;
    if (int rc = wait_for_action(c, spin, seq_word)) return rc;
    if (merge_reported_invalid(c)) return fail(c, NLC_ERR_STATE, "partials still marked invalid after the re-run");
  }
  std::memcpy(action_host, pin_act, na * sizeof(double));
  return NLC_OK;
  NLC_GUARD_END(c)
}
