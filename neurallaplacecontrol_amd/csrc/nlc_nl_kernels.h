// Kernel templates of the representation MLP + ILT + rollout (dataflow: top of kernels_nl.hip) and the launcher bodies
// for ONE hidden width h = 16 HT.  Each width is its own translation unit (kernels_nl.hip: h = 128, the harness's
// hidden_units; kernels_nl_h64.hip: the class default w_nl.py:72; kernels_nl_h256.hip: config.py's alternative) so
// the three sets of instantiations compile in parallel.
#pragma once
#include "nlc_device.h"
#include "nlc_envcost.h"
#include "nlc_kernels.h"
#include "nlc_rollout.h"

// NLC_ROLLOUT_LDS_TABS: nl_rollout_kernel reads the network's small tables from a workgroup copy in LDS (NlTabsLds, nlc_rollout.h)
#ifndef NLC_ROLLOUT_LDS_TABS
#define NLC_ROLLOUT_LDS_TABS 1
#endif
// NLC_ROLLOUT_LDS_W2: layer 2's whole matrix in LDS as well (154 KB).  Measured (round 4, K = 16384): 1.105 -> 1.146 ms -- the
// k loop's fragment reads are hidden behind its MFMAs either way, and the LDS reads of the other three waves are not free.  Off.
#ifndef NLC_ROLLOUT_LDS_W2
#define NLC_ROLLOUT_LDS_W2 0
#endif

namespace nlc {

// ------------------------------------------------------------------ T-step rollout (planner)
// LIN: fixed Talbot / Stehfest models (NlNetArgs::lin): two epilogue MFMAs per slot group, no Fourier prefactor
template <int HT, int NT3, bool LIN = false>
__global__ __launch_bounds__(256) void nl_rollout_kernel(const RolloutArgs a) {
  const NlNetArgs& n = a.net;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, c = lane & 15;
  const int64_t k = ((int64_t)blockIdx.x * 4 + wave) * 16 + c;
  const bool valid = k < a.K;
  const int64_t kc = valid ? k : a.K - 1;
  const int d = n.d;

  // lane (q) owns latent indices i0 = q and i1 = 4 + q: state dims, then the two GRU latents
  const int i0 = q, i1 = 4 + q;
  double x0 = 0.0, x1 = 0.0, m0 = 0.0, m1 = 0.0, s0 = 1.0, s1 = 1.0;
  const bool first_chunk = a.t_begin == 0, last_chunk = a.t_end == a.T;
  const int ep = (int)(kc / a.Kep);  // episode of this lane's sample (0 for the single planner)
  const int uoff = ep * a.T * a.nu;
  const double* st = first_chunk ? a.state0 + (a.state_per_sample ? kc : (int64_t)ep) * d : a.xcarry + kc * d;
  if (i0 < d) {
    x0 = st[i0];
    m0 = n.state_mean[i0];
    s0 = n.state_std[i0];
  }
  if (i1 < d) {
    x1 = st[i1];
    m1 = n.state_mean[i1];
    s1 = n.state_std[i1];
  }
  // ILT prefactor e^{gamma t}/T: constant over the rollout (ts_pred is constant)
  const double Tt = n.scale * a.tn;
  const double gamma = n.alpha - n.log_tol / (n.scale * Tt);
  const double factor = LIN ? 1.0 : exp(gamma * a.tn) / Tt;

  double cost = 0.0, pcost = 0.0;
  if (!first_chunk) {
    cost = a.ccarry[kc * 2];
    pcost = a.ccarry[kc * 2 + 1];
  }
  PhaseClk clk;
  clk.start();
  // One wave per SIMD has nobody to hide a load behind: everything a step reads that does not depend on the state -- the next
  // step's GRU latents, this step's action, noise and nominal control, the next evaluation's layer-1 tiles -- is issued one phase
  // or one step ahead (NLC_EVAL_PIPELINE, nlc_rollout.h).
#if NLC_ROLLOUT_LDS_TABS
  // (layer 2's matrix too where everything fits the CU's 160 KB)
  constexpr bool kW2 = NLC_ROLLOUT_LDS_W2 && NlTabsLds<HT, NT3, true>::kDoubles * 8 <= 160 * 1024;
  using Tabs = NlTabsLds<HT, NT3, kW2>;
  __shared__ double tabs_sm[Tabs::kDoubles];
  Tabs::fill(tabs_sm, n, threadIdx.x, 256);
  __syncthreads();
  const Tabs tabs{(lptr)tabs_sm};
#else
  const NlTabsGlobal tabs;
#endif
  // (hidden_units 256: layer 1 alone is 192 registers -- loaded where it is used)
  using Pre = typename std::conditional<(HT <= 8), NlL1Pre<HT>, NlNoPre>::type;
  Pre l1;
  if constexpr (HT <= 8) l1.load(n, lane, q, tabs);
  double pa0 = 0.0, pa1 = 0.0;
  if (a.t_begin < a.t_end) {
    const double* pa = a.pa + (kc * a.T + a.t_begin) * 2;
    pa0 = pa[0];
    pa1 = pa[1];
  }
  for (int t = a.t_begin; t < a.t_end; ++t) {
    double pert[NLC_MAX_NU] = {0.0, 0.0}, eps[NLC_MAX_NU] = {0.0, 0.0}, Ut[NLC_MAX_NU] = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < NLC_MAX_NU; ++j)
      if (j < a.nu) {
        pert[j] = a.perturbed[(kc * a.T + t) * a.nu + j];
        eps[j] = a.noise[(kc * a.T + t) * a.nu + j];
        Ut[j] = a.U[uoff + t * a.nu + j];
      }
    const double* pan = a.pa + (kc * a.T + (t + 1 < a.t_end ? t + 1 : t)) * 2;
    const double npa0 = pan[0], npa1 = pan[1];
    __builtin_amdgcn_sched_barrier(0);
    const double p0 = (i0 < d) ? (x0 - m0) / s0 : (i0 == d ? pa0 : (i0 == d + 1 ? pa1 : 0.0));
    const double p1 = (i1 < d) ? (x1 - m1) / s1 : (i1 == d ? pa0 : (i1 == d + 1 ? pa1 : 0.0));
    const v4d ax = nl_eval_impl<HT, NT3, false, false, LIN>(n, lane, q, p0, p1, a.tn, nullptr, nullptr, &clk, l1, tabs);
    // state + model(state, window, ts_pred)   (mppi_with_model.py:120-121)
    if (i0 < d) x0 = x0 + factor * ax[0];
    if (i1 < d) x1 = x1 + factor * ax[1];
    if (valid && a.states != nullptr) {
      double* so = a.states + (k * a.T + t) * d;
      if (i0 < d) so[i0] = x0;
      if (i1 < d) so[i1] = x1;
    }
    // gather the sample's full state into every lane of its column
    double xs[NLC_MAX_D];
#pragma unroll
    for (int i = 0; i < NLC_MAX_D; ++i) xs[i] = __shfl((i < 4) ? x0 : x1, ((i & 3) << 4) | c, 64);
    double u[NLC_MAX_NU] = {0.0, 0.0};
    double pc = 0.0;
#pragma unroll
    for (int j = 0; j < NLC_MAX_NU; ++j)
      if (j < a.nu) u[j] = a.u_scale * pert[j];
    // perturbation cost sum_j U[t,j] * (lambda * eps @ Sigma^-1)[j]   (mppi_delay.py:335,343)
#pragma unroll
    for (int j = 0; j < NLC_MAX_NU; ++j)
      if (j < a.nu) {
        double acj = 0.0;
#pragma unroll
        for (int i = 0; i < NLC_MAX_NU; ++i)
          if (i < a.nu) {
            double e = eps[i];
            if (a.noise_abs_cost) e = fabs(e);
            acj += (a.lambda_ * e) * a.sigma_inv[i * a.nu + j];
          }
        pc += Ut[j] * acj;
      }
    cost += running_cost(a.env, xs, u, a.nu);
    pcost += pc;
    pa0 = npa0;
    pa1 = npa1;
    clk.mark(PhaseClk::kTail);
  }
#if NLC_PHASE_CLOCKS
  if (lane == 0) {
    for (int i = 0; i < PhaseClk::kN; ++i) atomicAdd(&nlc_phase_clk[i], (unsigned long long)clk.acc[i]);
    atomicAdd(&nlc_phase_clk[PhaseClk::kN], 1ull);  // waves
  }
#endif
  if (valid) {
    if (last_chunk) {
      if (q == 0) a.cost_total[k] = cost + pcost;
    } else {
      if (i0 < d) a.xcarry[k * d + i0] = x0;
      if (i1 < d) a.xcarry[k * d + i1] = x1;
      if (q == 0) {
        a.ccarry[k * 2] = cost;
        a.ccarry[k * 2 + 1] = pcost;
      }
    }
  }
}

// ------------------------------------------------------------------ latency-split rollout (small K per GPU)
// one workgroup per 16-sample tile: rollout_split_tile (nlc_rollout.h), GRU latents from the (K, T, 2) tensor
template <int HT, int NT3, bool LIN = false>
__global__ __launch_bounds__(256) void nl_rollout_split_kernel(const RolloutArgs a) {
  constexpr int KS = HT * 4;
  __shared__ double H1[KS * 64], H2[KS * 64], AX[4 * 2 * 64];
  PaDirect src{a.pa, a.T, 0.0, 0.0, 0.0, 0.0};
  rollout_split_tile<HT, NT3, PaDirect, LIN>(a, (int64_t)blockIdx.x, src, H1, H2, AX);
}

// ------------------------------------------------------------------ single model forward, per-sample t
// GENERAL_T = false: every row shares one query time (ForwardArgs::tn; the constant sphere inputs of layer 1 are folded
// into net.b1 on the host, as for the planner) -- no W1s k-steps, no atan2 / asin per lane
template <int HT, int NT3, bool GENERAL_T = true>
__global__ __launch_bounds__(256) void nl_forward_kernel(const ForwardArgs a) {
  const NlNetArgs& n = a.net;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, c = lane & 15;
  const int64_t k = ((int64_t)blockIdx.x * 4 + wave) * 16 + c;
  const bool valid = k < a.N;
  const int64_t kc = valid ? k : a.N - 1;
  const int d = n.d;
  const int i0 = q, i1 = 4 + q;
  const double* ob = a.obs + kc * d;
  const double* pa = a.pa + kc * 2;
  const double p0 = (i0 < d) ? (ob[i0] - n.state_mean[i0]) / n.state_std[i0]
                             : (i0 == d ? pa[0] : (i0 == d + 1 ? pa[1] : 0.0));
  const double p1 = (i1 < d) ? (ob[i1] - n.state_mean[i1]) / n.state_std[i1]
                             : (i1 == d ? pa[0] : (i1 == d + 1 ? pa[1] : 0.0));
  const double tn = GENERAL_T ? a.ts[kc] / n.time_div : a.tn;  // w_nl.py:122
  const v4d ax = nl_eval<HT, NT3, GENERAL_T>(n, lane, q, p0, p1, tn);
  const double Tt = n.scale * tn;
  const double gamma = n.alpha - n.log_tol / (n.scale * Tt);
  const double factor = exp(gamma * tn) / Tt;
  if (valid) {
    if (i0 < d) a.out[k * d + i0] = factor * ax[0];
    if (i1 < d) a.out[k * d + i1] = factor * ax[1];
  }
}

// ------------------------------------------------------------------ representation function only (de Hoog path)
// One model evaluation per sample, output = F_k (re, im) of all d*S Laplace terms.  Used (a) per horizon step by
// the de Hoog planner path (folded constant-t bias) and (b) by NeuralLaplaceModel.forward with per-row t.
template <int HT, int NT3, bool GENERAL_T>
__global__ __launch_bounds__(256) void nl_repfunc_kernel(const RepFuncArgs a) {
  const NlNetArgs& n = a.net;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, c = lane & 15;
  const int64_t k = ((int64_t)blockIdx.x * 4 + wave) * 16 + c;
  const bool valid = k < a.N;
  const int64_t kc = valid ? k : a.N - 1;
  const int d = n.d;
  const int i0 = q, i1 = 4 + q;
  const double* ob = a.obs + (a.obs_per_sample ? kc : kc / a.Kep) * a.obs_stride;
  const double* pa = a.pa + kc * a.pa_stride;
  double x0 = (i0 < d) ? ob[i0] : 0.0, x1 = (i1 < d) ? ob[i1] : 0.0;
  if constexpr (!GENERAL_T) {
    if (a.tail_prev) {
      // tail of the previous horizon step (step_tail_kernel's arithmetic, in the MFMA lane layout: lane group q owns
      // state dims q and 4+q of sample c): x <- x + dx (mppi_with_model.py:120-121), store, running cost
      const StepTailArgs& s = a.tail;
      const int64_t e = kc / s.Kep;
      const double* src = s.first ? s.state0 + (s.state_per_sample ? kc : e) * d : s.x + kc * d;
      if (i0 < d) x0 = src[i0] + s.dx[kc * d + i0];
      if (i1 < d) x1 = src[i1] + s.dx[kc * d + i1];
      double xs[NLC_MAX_D];
#pragma unroll
      for (int i = 0; i < NLC_MAX_D; ++i) xs[i] = __shfl((i < 4) ? x0 : x1, ((i & 3) << 4) | c, 64);
      if (valid) {
        if (i0 < d) s.x[k * d + i0] = x0;
        if (i1 < d) s.x[k * d + i1] = x1;
        if (s.states != nullptr) {
          if (i0 < d) s.states[(k * s.T + s.t) * d + i0] = x0;
          if (i1 < d) s.states[(k * s.T + s.t) * d + i1] = x1;
        }
      }
      if (q == 0 && valid) {
        double u[NLC_MAX_NU] = {0.0, 0.0};
        for (int j = 0; j < s.nu; ++j) u[j] = s.u_scale * s.perturbed[(k * s.T + s.t) * s.nu + j];
        const double pc = perturbation_cost_step(s.noise + (k * s.T + s.t) * s.nu, s.U + (e * s.T + s.t) * s.nu, s.sigma_inv,
                                                 s.lambda_, s.nu, s.noise_abs_cost);
        s.ccarry[k * 2] = (s.first ? 0.0 : s.ccarry[k * 2]) + running_cost(s.env, xs, u, s.nu);
        s.ccarry[k * 2 + 1] = (s.first ? 0.0 : s.ccarry[k * 2 + 1]) + pc;
      }
    }
  }
  const double p0 = (i0 < d) ? (x0 - n.state_mean[i0]) / n.state_std[i0]
                             : (i0 == d ? pa[0] : (i0 == d + 1 ? pa[1] : 0.0));
  const double p1 = (i1 < d) ? (x1 - n.state_mean[i1]) / n.state_std[i1]
                             : (i1 == d ? pa[0] : (i1 == d + 1 ? pa[1] : 0.0));
  const FOut fo{a.fre, a.fim, a.slot, valid ? k : -1, d * n.S, a.write_angles, a.slot_major ? a.N : (int64_t)0};
  if constexpr (GENERAL_T) {
    const double* sph_row = a.sph != nullptr ? a.sph + kc * a.sph_stride : nullptr;
    const double tn = a.sph != nullptr ? 1.0 : a.ts[kc] / n.time_div;
    nl_eval<HT, NT3, true, true>(n, lane, q, p0, p1, tn, &fo, sph_row);
  } else {
    nl_eval<HT, NT3, false, true>(n, lane, q, p0, p1, a.tn, &fo);
  }
}

// latency-split form of the planner's per-step launch (repfunc_split_tile, nlc_rollout.h)
template <int HT, int NT3>
__global__ __launch_bounds__(256, 2) void nl_repfunc_split_kernel(const RepFuncArgs a) {
  constexpr int KS = HT * 4;
  __shared__ double H1[KS * 64], H2[KS * 64];
  repfunc_split_tile<HT, NT3>(a, (int64_t)blockIdx.x, H1, H2);
}

// instantiated layer-3 tile counts; other (d,S) round up to the next one (zero-padded tiles)
#define NLC_FOR_NT3(X) X(7) X(9) X(11) X(13) X(17) X(21) X(25)

}  // namespace nlc
