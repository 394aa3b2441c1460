// MPPI sampling, bounding, importance weighting and the oracle-dynamics rollout.
//   MPPIDelay.command                 planners/mppi_delay.py:193-224
//   MPPIDelay._compute_total_cost_batch  :315-345   (_bound_action :347-353)
//   oracle.*_dynamics_dt_delay        oracle.py:11-224        (rollout with model_name == "oracle")
// All of these are HBM-light streaming / reduction kernels over the (K, T, nu) noise tensor.
#include "nlc_device.h"
#include "nlc_kernels.h"
#include "nlc_mppi_dev.h"

namespace nlc {

// ------------------------------------------------------------------ U <- roll(U, -1); U[-1] = u_init  (:199-200)
// blockIdx.x = episode
__global__ void shift_U_kernel(const PerturbArgs a) {
  const int n = a.T * a.nu;
  const double* Uo = a.U_old + (int64_t)blockIdx.x * n;
  double* Un = a.U_new + (int64_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int t = i / a.nu, j = i - t * a.nu;
    Un[i] = (t + 1 < a.T) ? Uo[i + a.nu] : a.u_init[j];
  }
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < a.n_state_in; i += blockDim.x) a.state_dst[i] = a.state_in[i];
    for (int i = threadIdx.x; i < a.n_abuf_in; i += blockDim.x) a.abuf_dst[i] = a.abuf_in[i];
  }
}
hipError_t launch_shift_U(const PerturbArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(shift_U_kernel, dim3((unsigned)a.E), dim3(128), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ sample / perturb / bound (:319-328)
// One thread per (k, t).  rng == 1 draws eps ~ N(mu, Sigma) on the device: Philox4x32-10 keyed by the
// seed, counter (global sample index, t, command counter) -> the draw does not depend on the sharding.
// Episode e of a batched planner continues the index space at e * K_global (e == 0: the single planner's stream).
__global__ __launch_bounds__(256) void perturb_kernel(const PerturbArgs a) {
  const int64_t total = a.K * a.T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_zero_words; i += (int64_t)gridDim.x * blockDim.x)
    a.zero_words[i] = 0u;
  if (a.fused_shift && blockIdx.x == 0) {
    // the staged per-command inputs (shift_U_kernel's second job)
    for (int i = threadIdx.x; i < a.n_state_in; i += blockDim.x) a.state_dst[i] = a.state_in[i];
    for (int i = threadIdx.x; i < a.n_abuf_in; i += blockDim.x) a.abuf_dst[i] = a.abuf_in[i];
  }
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = idx / a.T;
    const int t = (int)(idx - k * a.T);
    const int64_t e = k / a.Kep;
    const int64_t ke = a.k_offset + (k - e * a.Kep);  // index within the episode's whole population
    const int64_t kg = e * a.K_global + ke;
    double eps[NLC_MAX_NU];
    if (a.rng) {
      mppi_draw(kg, t, a.seed, a.counter, a.nu, a.mu, a.chol, eps);
    } else {
      for (int i = 0; i < a.nu; ++i) eps[i] = a.noise[idx * a.nu + i];
    }
    const bool null_action = a.sample_null_action && (ke == a.K_global - 1);
    for (int i = 0; i < a.nu; ++i) {
      double U;
      if (a.fused_shift) {
        // U <- roll(U, -1); U[-1] = u_init (:199-200) applied on the fly; the episode's first local sample stores it
        U = (t + 1 < a.T) ? a.U_old[(e * a.T + t + 1) * a.nu + i] : a.u_init[i];
        if (k == e * a.Kep) a.U_new[(e * a.T + t) * a.nu + i] = U;
      } else {
        U = a.U_new[(e * a.T + t) * a.nu + i];
      }
      const double V = mppi_bound(U, eps[i], null_action, a.u_scale, a.has_bounds, a.u_min[i], a.u_max[i]);
      a.perturbed[idx * a.nu + i] = V;
      a.noise[idx * a.nu + i] = V - U;                               // :328
      if (a.actions != nullptr) a.actions[idx * a.nu + i] = (a.u_scale * V) / a.u_scale;  // :255,340
    }
  }
}
hipError_t launch_perturb(const PerturbArgs& a, hipStream_t s) {
  const int64_t total = a.K * a.T;
  if (total <= 0) return hipSuccess;
  const int64_t want = (total + 255) / 256;
  const unsigned grid = (unsigned)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(perturb_kernel, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ importance weights (:210-216)
// Two launches (arithmetic and its fixed order: nlc_mppi_dev.h): one wavefront per 16-sample tile, then one workgroup per
// episode folds the tile partials into the shard's (beta_r, eta_r, S_r).  blockIdx.y = episode.
__global__ __launch_bounds__(256) void weight_tile_kernel(const WeightArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.nblk) return;
  const int64_t e = blockIdx.y;
  const int64_t ks = b * kWeightTile + (lane & 15);  // within the episode
  const bool valid = ks < a.Kep;
  const double cost = valid ? a.cost[e * a.Kep + ks] : 0.0;
  weight_tile<MemPlain>(a, e, b, lane, cost, valid);
}
// small populations: one workgroup per episode folds all chunks (weight_rank)
__global__ __launch_bounds__(256) void weight_rank_kernel(const WeightArgs a) {
  __shared__ double lds[kWeightRankLds];
  weight_rank<MemPlain>(a, (int64_t)blockIdx.y, lds);
  // (thread 0 stored eta_r) the fused launch before this fold gave up: mark the row, every rank sees it after the all-gather
  if (a.gave_up != nullptr && threadIdx.x == 0 && __hip_atomic_load(a.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
    a.partials[(int64_t)blockIdx.y * (2 + a.T * a.nu) + 1] = kPartialInvalidEta;
}
// large populations: one workgroup per 64-tile chunk, then one per episode adds the chunk values in ascending order --
// the same additions in the same order as weight_rank
__global__ __launch_bounds__(256) void weight_chunk_kernel(const WeightArgs a) {
  __shared__ double lds[kWeightRankLds];
  const int64_t e = blockIdx.y;
  const int j = blockIdx.x, lane = threadIdx.x & 63, TN = a.T * a.nu;
  const double beta = weight_beta<MemPlain>(a, e, lds);
  for (int i0 = 0; i0 < 1 + TN; i0 += 64) {
    const double c = weight_chunk<MemPlain>(a, e, j, i0, beta, lds);
    if (threadIdx.x < 64 && i0 + lane < 1 + TN) a.chunk_part[(e * gridDim.x + j) * (2 + TN) + 1 + i0 + lane] = c;
  }
  if (threadIdx.x == 0) a.chunk_part[(e * gridDim.x + j) * (2 + TN)] = beta;
}
__global__ __launch_bounds__(128) void weight_final_kernel(const WeightArgs a, int nch) {
#pragma clang fp contract(off)
  const int64_t e = blockIdx.y;
  const int TN = a.T * a.nu, W = 2 + TN;
  const double* cp = a.chunk_part + e * nch * W;
  for (int i = threadIdx.x; i < 1 + TN; i += 128) {
    double tot = 0.0;
    for (int j0 = 0; j0 < nch; j0 += 16) {
      double v[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) v[s] = j0 + s < nch ? cp[(int64_t)(j0 + s) * W + 1 + i] : 0.0;
#pragma unroll
      for (int s = 0; s < 16; ++s)
        if (j0 + s < nch) tot += v[s];
    }
    a.partials[e * W + 1 + i] = tot;
  }
  if (threadIdx.x == 0) {
    a.partials[e * W] = cp[0];
    // (this thread stored eta_r above: i = 0) see weight_rank_kernel
    if (a.gave_up != nullptr && __hip_atomic_load(a.gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
      a.partials[e * W + 1] = kPartialInvalidEta;
  }
}

hipError_t launch_weights(const WeightArgs& a, hipStream_t s) {
  const unsigned E = (unsigned)a.E;
  hipLaunchKernelGGL(weight_tile_kernel, dim3((a.nblk + 3) / 4, E), dim3(256), 0, s, a);
  const int nch = weight_chunks(a.nblk);
  if (nch <= 4) {
    hipLaunchKernelGGL(weight_rank_kernel, dim3(1, E), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(weight_chunk_kernel, dim3(nch, E), dim3(256), 0, s, a);
    hipLaunchKernelGGL(weight_final_kernel, dim3(1, E), dim3(128), 0, s, a, nch);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------ merge shard partials, update U (:213-224)
// gathered[g] = (beta_g, eta_g, S_g[T*nu]).  beta = min beta_g, scale_g = exp(-(beta_g - beta)/lambda),
// eta = sum scale_g eta_g, dU = sum scale_g S_g / eta   (SURVEY §8e).  Every rank runs this identically.
__global__ __launch_bounds__(256) void merge_kernel(const MergeArgs a) {
  __shared__ double s_scale_self, s_eta;
  const int TN = a.T * a.nu;
  const int W = 2 + TN;
  const int64_t e = blockIdx.y;
  // the fused planner body of the NEXT command starts from zeroed tickets / flags (this launch is the last of a command)
  for (int64_t i = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < a.n_zero_words;
       i += (int64_t)gridDim.x * gridDim.y * 256)
    a.zero_words[i] = 0u;
  const double* gat = a.gathered + e * W;       // rank g's row of this episode: gat + g * E * W
  const int64_t gs = (int64_t)a.E * W;
  // a shard whose fused launch gave up marks its row (kPartialInvalidEta): identical on every rank after the all-gather, so
  // every rank skips the update, reports it to its host and the command is re-run everywhere (nlc_mppi_finish)
  bool invalid = false;
  for (int g = 0; g < a.G; ++g) invalid = invalid || gat[g * gs + 1] < 0.0;
  if (invalid) {
    if (blockIdx.x == 0) {
      const double nan = __builtin_nan("");
      for (int i = threadIdx.x; i < a.u_per_command * a.nu; i += 256) {
        a.action[e * a.u_per_command * a.nu + i] = nan;
        if (a.action_pinned != nullptr) a.action_pinned[e * a.u_per_command * a.nu + i] = nan;
      }
      if (a.status_pinned != nullptr && threadIdx.x == 0)
        __hip_atomic_store(a.status_pinned, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (a.seq_pinned != nullptr) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(a.seq_pinned, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
  double beta = INFINITY;
  for (int g = 0; g < a.G; ++g) beta = fmin(beta, gat[g * gs]);
  double eta = 0.0;
  for (int g = 0; g < a.G; ++g) eta += exp(-(gat[g * gs] - beta) / a.lambda_) * gat[g * gs + 1];
  if (blockIdx.x == 0) {
    double* U = a.U + e * TN;
    for (int i = threadIdx.x; i < TN; i += 256) {
      double acc = 0.0;
      for (int g = 0; g < a.G; ++g) acc += exp(-(gat[g * gs] - beta) / a.lambda_) * gat[g * gs + 2 + i];
      const double u = U[i] + (1.0 / eta) * acc;  // omega = (1/eta) w, :214-216
      U[i] = u;
      if (i < a.u_per_command * a.nu) {
        a.action[e * a.u_per_command * a.nu + i] = u * a.u_scale;  // :217-224
        if (a.action_pinned != nullptr) a.action_pinned[e * a.u_per_command * a.nu + i] = u * a.u_scale;
      }
    }
    if (threadIdx.x == 0) {
      a.beta_eta[e * 2] = beta;
      a.beta_eta[e * 2 + 1] = eta;
    }
    if (a.seq_pinned != nullptr) {
      // the action words above have left this workgroup (every thread drains its stores, then the barrier); ONE
      // system-scope store tells the spinning host that they are in its memory
      __threadfence_system();
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(a.seq_pinned, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // cost_total_non_zero = exp(-(c - beta)/lambda) with the GLOBAL beta (_ensure_non_zero :12-13, :213) and omega (:214)
  if (threadIdx.x == 0) s_eta = eta;
  (void)s_scale_self;
  __syncthreads();
  const double inv = 1.0 / s_eta;
  const double* cost = a.cost + e * a.Kep;
  double* cost_nz = a.cost_nz + e * a.Kep;
  double* omega = a.omega != nullptr ? a.omega + e * a.Kep : nullptr;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < a.Kep; k += (int64_t)gridDim.x * 256) {
    const double w = exp(-(1.0 / a.lambda_) * (cost[k] - beta));
    cost_nz[k] = w;
    if (omega != nullptr) omega[k] = inv * w;
  }
}
hipError_t launch_merge(const MergeArgs& a, hipStream_t s) {
  const int64_t want = (a.Kep + 255) / 256;
  const unsigned grid = (unsigned)(want < 1024 ? (want > 0 ? want : 1) : 1024);
  hipLaunchKernelGGL(merge_kernel, dim3(grid, (unsigned)a.E), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ oracle-dynamics rollout
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

__device__ __forceinline__ double trig2angle_o(double c, double s) {
  const double C = c * c + s * s;
  c = c / C;
  s = s / C;
  return atan2(s / C, c / C);
}

// running costs: same formulas as kernels_nl.hip (kept local: this TU has no MFMA code)
__device__ __forceinline__ double running_cost_o(int env, const double* x, const double* u, int nu) {
  if (env < 0) return 0.0;  // cost_external: the caller evaluates its own running cost on the stored states
  double uu = 0.0;
  for (int j = 0; j < nu; ++j) uu += u[j] * u[j];
  if (env == NLC_ENV_CARTPOLE) {
    const double e0 = x[0] + x[3] - 0.0, e1 = x[2] - 1.0;
    const double sr = -(e0 * e0 + e1 * e1);
    const double vr = -(x[1] * x[1]) - x[4] * x[4];
    return -((sr + 0.01 * vr) + (-0.01 * uu));
  } else if (env == NLC_ENV_CARTPOLE_NOTRIG) {  // ctcartpole.py:297-300: explicit angle
    const double cl = 1.0 * cos(x[2]), sl = 1.0 * sin(x[2]);
    const double e0 = x[0] + sl - 0.0, e1 = cl - 1.0;
    const double sr = -(e0 * e0 + e1 * e1);
    const double vr = -(x[1] * x[1]) - x[3] * x[3];
    return -((sr + 0.01 * vr) + (-0.01 * uu));
  } else if (env == NLC_ENV_PENDULUM) {
    const double om = 1.0 - x[0];
    const double sr = -(om * om + x[1] * x[1]);
    const double vr = -(x[2] * x[2]);
    return -((sr + 0.01 * vr) + (-0.01 * uu));
  } else {
    const double th1 = trig2angle_o(x[0], x[1]), th2 = trig2angle_o(x[2], x[3]);
    const double vr = -(x[4] * x[4]) - x[5] * x[5];
    const double p1x = -cos(th1), p1y = sin(th1);
    const double p2x = p1x - cos(th1 + th2), p2y = p1y + sin(th1 + th2);
    const double ex = p2x - 1.0 - 1.0;
    const double sr = -(ex * ex) - p2y * p2y;
    return -((sr + 1e-1 * vr) + (-1e-4 * uu));
  }
}

// One Euler step of the closed-form dynamics on the trig observation (oracle.py:11-86, 89-174, 177-224)
__device__ __forceinline__ void oracle_step(int env, double* x, const double* uraw, double ts, int friction) {
  if (env == NLC_ENV_CARTPOLE || env == NLC_ENV_CARTPOLE_NOTRIG) {
    const bool trig = env == NLC_ENV_CARTPOLE;  // oracle.py:29-37 (5-D observation) / :38-44 (4-D state: explicit angle)
    const double u = clampd(uraw[0], -3.0, 3.0);
    double xx = x[0], xd = x[1], c, s, thd, th;
    if (trig) {
      c = x[2];
      s = x[3];
      thd = x[4];
      const double C = c * c + s * s;
      c = c / C;
      s = s / C;
      th = atan2(s / C, c / C);
    } else {
      th = x[2];
      thd = x[3];
      c = cos(th);
      s = sin(th);
    }
    const double g = 9.8, fmag = 3.0, mc = 1.0, mp = 0.1, len = 1.0;
    const double mt = mp + mc, pml = mp * len;
    const double force = u * fmag;
    double temp, thacc;
    if (friction) {
      const double sg = (xd > 0.0) ? 1.0 : ((xd < 0.0) ? -1.0 : 0.0);
      temp = (force + pml * thd * thd * s - 5e-4 * sg) / mt;
      thacc = (g * s - c * temp - 2e-6 * thd / pml) / (len * (4.0 / 3.0 - mp * c * c / mt));
    } else {
      temp = (force + pml * thd * thd * s) / mt;
      thacc = (g * s - c * temp) / (len * (4.0 / 3.0 - mp * c * c / mt));
    }
    const double xacc = temp - pml * thacc * c / mt;
    const double nthd = thd + thacc * ts, nth = th + thd * ts;
    const double nxd = xd + xacc * ts, nx = xx + xd * ts;
    x[0] = nx;
    x[1] = nxd;
    if (trig) {
      x[2] = cos(nth);
      x[3] = sin(nth);
      x[4] = nthd;
    } else {  // oracle.py:80-86
      x[2] = nth;
      x[3] = nthd;
    }
  } else if (env == NLC_ENV_PENDULUM) {
    const double u = clampd(uraw[0], -2.0, 2.0);
    const double c = x[0], s = x[1], thd = x[2];
    const double C = c * c + s * s;
    const double th = atan2((s / C) / C, (c / C) / C);
    const double nth = th + thd * ts;
    const double nthd = thd + (-15.0 * sin(th + kPi) + 3.0 * u) * ts;  // -3g/(2l), 3/(m l^2)
    x[0] = cos(nth);
    x[1] = sin(nth);
    x[2] = nthd;
  } else {
    const double u0 = clampd(uraw[0], -5.0, 5.0), u1 = clampd(uraw[1], -5.0, 5.0);
    const double th1 = trig2angle_o(x[0], x[1]), th2 = trig2angle_o(x[2], x[3]);
    const double d1v = x[4], d2v = x[5];
    const double m1 = 1.0, m2 = 1.0, l1 = 1.0, lc1 = 0.5, lc2 = 0.5, I1 = 1.0, I2 = 1.0, g = 9.8;
    const double c2 = cos(th2), s2 = sin(th2);
    const double D1 = m1 * (lc1 * lc1) + m2 * (l1 * l1 + lc2 * lc2 + 2 * l1 * lc2 * c2) + I1 + I2;
    const double D2 = m2 * (lc2 * lc2 + l1 * lc2 * c2) + I2;
    const double phi2 = m2 * lc2 * g * cos(th1 + th2 - kPi / 2.0);
    const double phi1 = -m2 * l1 * lc2 * (d2v * d2v) * s2 - 2 * m2 * l1 * lc2 * d2v * d1v * s2 +
                        (m1 * lc1 + m2 * l1) * g * cos(th1 - kPi / 2) + phi2;
    const double dd2 = (u0 + D2 / D1 * phi1 - m2 * l1 * lc2 * (d1v * d1v) * s2 - phi2) /
                       (m2 * (lc2 * lc2) + I2 - (D2 * D2) / D1);
    const double dd1 = -(u1 + D2 * dd2 + phi1) / D1;
    const double nd1 = d1v + dd1 * ts, nd2 = d2v + dd2 * ts;
    const double nth1 = th1 + d1v * ts, nth2 = th2 + d2v * ts;
    x[0] = cos(nth1);
    x[1] = sin(nth1);
    x[2] = cos(nth2);
    x[3] = sin(nth2);
    x[4] = nd1;
    x[5] = nd2;
  }
}

__global__ __launch_bounds__(256) void oracle_rollout_kernel(const OracleRolloutArgs a) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= a.K) return;
  double x[NLC_MAX_D];
  const int64_t e = k / a.Kep;
  const double* st = a.state0 + (a.state_per_sample ? k : e) * a.d;
  const double* abuf = a.abuf + e * a.B * a.nu;
  const double* U = a.U + e * a.T * a.nu;
  for (int i = 0; i < a.d; ++i) x[i] = st[i];
  double cost = 0.0, pcost = 0.0;
  for (int t = 0; t < a.T; ++t) {
    // window (k,t) = hist[k][t : t+B]; applied action = window[-(delay+1)] = hist[k][t + B-1-delay]
    double ud[NLC_MAX_NU], u[NLC_MAX_NU];
    const int i = t + a.B - 1 - a.delay;
    for (int j = 0; j < a.nu; ++j) {
      ud[j] = (i < a.B - 1) ? abuf[(1 + i) * a.nu + j]
                            : a.u_scale * a.perturbed[(k * a.T + (i - (a.B - 1))) * a.nu + j];
      u[j] = a.u_scale * a.perturbed[(k * a.T + t) * a.nu + j];
    }
    oracle_step(a.env, x, ud, a.ts, a.friction);
    if (a.states != nullptr)
      for (int ii = 0; ii < a.d; ++ii) a.states[(k * a.T + t) * a.d + ii] = x[ii];
    double pc = 0.0;
    for (int j = 0; j < a.nu; ++j) {
      double acj = 0.0;
      for (int ii = 0; ii < a.nu; ++ii) {
        double e = a.noise[(k * a.T + t) * a.nu + ii];
        if (a.noise_abs_cost) e = fabs(e);
        acj += (a.lambda_ * e) * a.sigma_inv[ii * a.nu + j];
      }
      pc += U[t * a.nu + j] * acj;
    }
    cost += running_cost_o(a.cost_env, x, u, a.nu);
    pcost += pc;
  }
  a.cost_total[k] = cost + pcost;
}
hipError_t launch_oracle_rollout(const OracleRolloutArgs& a, hipStream_t s) {
  if (a.K <= 0) return hipSuccess;
  hipLaunchKernelGGL(oracle_rollout_kernel, dim3((unsigned)((a.K + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ env side of the evaluation loop
// Reduced-state rhs of the three envs (ctcartpole.py:185-237, ctpendulum.py:111-125, ctacrobot.py:168-228; the 4-D /
// 2-D branches: explicit angles).  Only cartpole clamps the action inside its rhs.
__device__ __forceinline__ void env_rhs(int env, const double* s, const double* a, int friction, double* ds) {
  if (env == NLC_ENV_CARTPOLE) {
    const double xd = s[1], th = s[2], thd = s[3];
    const double c = cos(th), sn = sin(th);
    const double g = 9.8, fmag = 3.0, mc = 1.0, mp = 0.1, len = 1.0;
    const double mt = mp + mc, pml = mp * len;
    const double force = clampd(a[0], -fmag, fmag) * fmag;
    double temp, thacc;
    if (friction) {
      const double sg = (xd > 0.0) ? 1.0 : ((xd < 0.0) ? -1.0 : 0.0);
      temp = (force + pml * thd * thd * sn - 5e-4 * sg) / mt;
      thacc = (g * sn - c * temp - 2e-6 * thd / pml) / (len * (4.0 / 3.0 - mp * c * c / mt));
    } else {
      temp = (force + pml * thd * thd * sn) / mt;
      thacc = (g * sn - c * temp) / (len * (4.0 / 3.0 - mp * c * c / mt));
    }
    ds[0] = xd;
    ds[1] = temp - pml * thacc * c / mt;
    ds[2] = thd;
    ds[3] = thacc;
  } else if (env == NLC_ENV_PENDULUM) {
    ds[0] = s[1];
    ds[1] = -15.0 * sin(s[0] + kPi) + 3.0 * a[0];  // -3g/(2l), 3/(m l^2)
  } else {
    const double th1 = s[0], th2 = s[1], d1v = s[2], d2v = s[3];
    const double m1 = 1.0, m2 = 1.0, l1 = 1.0, lc1 = 0.5, lc2 = 0.5, I1 = 1.0, I2 = 1.0, g = 9.8;
    const double c2 = cos(th2), s2 = sin(th2);
    const double D1 = m1 * (lc1 * lc1) + m2 * (l1 * l1 + lc2 * lc2 + 2 * l1 * lc2 * c2) + I1 + I2;
    const double D2 = m2 * (lc2 * lc2 + l1 * lc2 * c2) + I2;
    const double phi2 = m2 * lc2 * g * cos(th1 + th2 - kPi / 2.0);
    const double phi1 = -m2 * l1 * lc2 * (d2v * d2v) * s2 - 2 * m2 * l1 * lc2 * d2v * d1v * s2 +
                        (m1 * lc1 + m2 * l1) * g * cos(th1 - kPi / 2) + phi2;
    const double dd2 = (a[0] + D2 / D1 * phi1 - m2 * l1 * lc2 * (d1v * d1v) * s2 - phi2) /
                       (m2 * (lc2 * lc2) + I2 - (D2 * D2) / D1);
    ds[0] = d1v;
    ds[1] = d2v;
    ds[2] = -(a[1] + D2 * dd2 + phi1) / D1;
    ds[3] = dd2;
  }
}
// torch_transform_states: reduced state -> trig observation
__device__ __forceinline__ void env_observe(int env, const double* s, double* o) {
  if (env == NLC_ENV_CARTPOLE) {
    o[0] = s[0];
    o[1] = s[1];
    o[2] = 1.0 * cos(s[2]);
    o[3] = 1.0 * sin(s[2]);
    o[4] = s[3];
  } else if (env == NLC_ENV_PENDULUM) {
    o[0] = cos(s[0]);
    o[1] = sin(s[0]);
    o[2] = s[1];
  } else {
    o[0] = cos(s[0]);
    o[1] = sin(s[0]);
    o[2] = cos(s[1]);
    o[3] = sin(s[1]);
    o[4] = s[2];
    o[5] = s[3];
  }
}
// diff_reward(s, a) on the reduced state, as integrate_system evaluates it (base_env.py:164)
__device__ __forceinline__ double env_reward(int env, const double* s, const double* a, int nu) {
  double uu = 0.0;
  for (int j = 0; j < nu; ++j) uu += a[j] * a[j];
  if (env == NLC_ENV_CARTPOLE) {
    const double e0 = s[0] + 1.0 * sin(s[2]) - 0.0, e1 = 1.0 * cos(s[2]) - 1.0;
    return (-(e0 * e0 + e1 * e1) + 0.01 * (-(s[1] * s[1]) - s[3] * s[3])) + (-0.01 * uu);
  } else if (env == NLC_ENV_PENDULUM) {
    const double c = cos(s[0]), sn = sin(s[0]);
    const double om = 1.0 - c;
    return (-1.0 * (om * om + sn * sn) + 0.01 * (-(s[1] * s[1]))) + (-0.01 * uu);
  }
  const double p1x = -1.0 * cos(s[0]), p1y = 1.0 * sin(s[0]);
  const double p2x = p1x - 1.0 * cos(s[0] + s[1]), p2y = p1y + 1.0 * sin(s[0] + s[1]);
  const double ex = p2x - 1.0 - 1.0;
  return ((-(ex * ex) - p2y * p2y) + 1e-1 * (-(s[2] * s[2]) - s[3] * s[3])) + (-1e-4 * uu);
}
__global__ __launch_bounds__(256) void env_step_kernel(const EnvStepArgs a) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= a.E) return;
  const int n = (a.env == NLC_ENV_PENDULUM) ? 2 : 4;
  const int d = (a.env == NLC_ENV_CARTPOLE) ? 5 : ((a.env == NLC_ENV_PENDULUM) ? 3 : 6);
  double s[4], o[6];
  for (int i = 0; i < n; ++i) s[i] = a.state[e * n + i];
  if (a.action != nullptr) {
    // get_action: roll the buffer by one row, append the new action, apply row -(delay + 1)
    double* ab = a.abuf + e * a.B * a.nu;
    for (int r = 0; r + 1 < a.B; ++r)
      for (int j = 0; j < a.nu; ++j) ab[r * a.nu + j] = ab[(r + 1) * a.nu + j];
    for (int j = 0; j < a.nu; ++j) ab[(a.B - 1) * a.nu + j] = a.action[e * a.nu + j];
    double at[NLC_MAX_NU], ds[4];
    for (int j = 0; j < a.nu; ++j) at[j] = ab[(a.B - 1 - a.delay) * a.nu + j];
    env_rhs(a.env, s, at, a.friction, ds);
    for (int i = 0; i < n; ++i) {
      s[i] = s[i] + a.dt * ds[i];  // odeint(method="euler") over ts = [0, dt]
      a.state[e * n + i] = s[i];
    }
    if (a.reward != nullptr) a.reward[e] = env_reward(a.env, s, at, a.nu);
  }
  env_observe(a.env, s, o);
  for (int i = 0; i < d; ++i) a.obs[e * d + i] = o[i];
}
hipError_t launch_env_step(const EnvStepArgs& a, hipStream_t s) {
  if (a.E <= 0) return hipSuccess;
  hipLaunchKernelGGL(env_step_kernel, dim3((unsigned)((a.E + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ Delta-t RNN baseline: linear_out tail + rollout
// dx = q + W_out[:, H:H+d] obs_n + W_out[:, H+d] ts_n + b   (train_utils.py:618-631), q = hidden part from rnn_encode_kernel
__device__ __forceinline__ void rnn_dx(const RnnHead& h, const double (&x)[NLC_MAX_D], const double* q, double tsn,
                                       double (&dx)[NLC_MAX_D]) {
  double on[NLC_MAX_D];
  for (int j = 0; j < h.d; ++j) on[j] = (x[j] - h.mean[j]) / h.std[j];
  for (int i = 0; i < h.d; ++i) {
    double acc = q[i];
    for (int j = 0; j < h.d; ++j) acc += h.Wx[i * h.d + j] * on[j];
    dx[i] = acc + h.wt[i] * tsn + h.b[i];
  }
}
__global__ __launch_bounds__(256) void rnn_forward_tail_kernel(const RnnForwardArgs a) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= a.N) return;
  const int d = a.head.d;
  double x[NLC_MAX_D], dx[NLC_MAX_D];
  for (int i = 0; i < d; ++i) x[i] = a.obs[n * d + i];
  rnn_dx(a.head, x, a.q + n * d, a.ts[n] / a.head.time_div, dx);
  for (int i = 0; i < d; ++i) a.out[n * d + i] = dx[i];
}
hipError_t launch_rnn_forward_tail(const RnnForwardArgs& a, hipStream_t s) {
  if (a.N <= 0) return hipSuccess;
  hipLaunchKernelGGL(rnn_forward_tail_kernel, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

// T-step rollout with the harness closure state + model(state, window, ts_pred) (mppi_with_model.py:103-122): one
// thread per sample; the GRU half of the model is already in q (T, K, d), so a step is a d x d matvec + the cost.
__global__ __launch_bounds__(256) void rnn_rollout_kernel(const RnnRolloutArgs a) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= a.K) return;
  const int d = a.head.d;
  double x[NLC_MAX_D], dx[NLC_MAX_D];
  const int64_t e = k / a.Kep;
  const double* st = a.state0 + (a.state_per_sample ? k : e) * d;
  const double* U = a.U + e * a.T * a.nu;
  for (int i = 0; i < d; ++i) x[i] = st[i];
  const double tsn = a.ts / a.head.time_div;
  double cost = 0.0, pcost = 0.0;
  for (int t = 0; t < a.T; ++t) {
    rnn_dx(a.head, x, a.q + ((int64_t)t * a.K + k) * d, tsn, dx);
    for (int i = 0; i < d; ++i) x[i] = x[i] + dx[i];
    double u[NLC_MAX_NU];
    for (int j = 0; j < a.nu; ++j) u[j] = a.u_scale * a.perturbed[(k * a.T + t) * a.nu + j];
    if (a.states != nullptr)
      for (int i = 0; i < d; ++i) a.states[(k * a.T + t) * d + i] = x[i];
    double pc = 0.0;
    for (int j = 0; j < a.nu; ++j) {
      double acj = 0.0;
      for (int ii = 0; ii < a.nu; ++ii) {
        double e2 = a.noise[(k * a.T + t) * a.nu + ii];
        if (a.noise_abs_cost) e2 = fabs(e2);
        acj += (a.lambda_ * e2) * a.sigma_inv[ii * a.nu + j];
      }
      pc += U[t * a.nu + j] * acj;
    }
    cost += running_cost_o(a.env, x, u, a.nu);
    pcost += pc;
  }
  a.cost_total[k] = cost + pcost;
}
hipError_t launch_rnn_rollout(const RnnRolloutArgs& a, hipStream_t s) {
  if (a.K <= 0) return hipSuccess;
  hipLaunchKernelGGL(rnn_rollout_kernel, dim3((unsigned)((a.K + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace nlc
