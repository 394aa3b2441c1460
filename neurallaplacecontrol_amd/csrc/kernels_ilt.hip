// Stand-alone inverse-Laplace-transform kernels: torchlaplace.laplace_reconstruct around an arbitrary
// representation function (external package; reference call sites w_nl.py:137-144, w_latent_ode.py:88-94).
//
//   rep_inputs_kernel   contour evaluation s_k(t) = gamma + i pi k/T, Riemann-sphere projection
//                       (theta_s, phi_s) and concatenation with the latent p.
//   ilt_fourier_kernel  sphere -> complex F_k = tan(phi/2+pi/4) e^{i theta} and the Fourier-series line
//                       integral.  HBM-bound stream: reads (2 d S) f64, writes d f64 per point
//                       ((2dS+d)*8 algorithmic bytes; 1400 B at d=5, S=17).
//   ilt_dehoog_kernel   same map, de Hoog-Knight-Stokes quotient-difference acceleration
//                       (mpmath inverselaplace.py:476-531); O(M^2) complex ops per (point, dim) -> FP64
//                       VALU bound, not HBM bound.
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "nlc_device.h"
#include "nlc_kernels.h"

#ifndef NLC_ILT_EXPERIMENTS
#define NLC_ILT_EXPERIMENTS 0
#endif

namespace nlc {

// ------------------------------------------------------------------ rep-func inputs
__global__ __launch_bounds__(256) void rep_inputs_kernel(const RepInArgs a) {
  const int W = 2 * a.S + a.P;
  const int64_t total = a.B * a.Tt * W;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / W;
    const int col = (int)(idx - row * W);
    const int64_t b = row / a.Tt;
    const int64_t j = row - b * a.Tt;
    double v;
    if (col >= 2 * a.S) {
      v = a.p[b * a.P + (col - 2 * a.S)];
    } else {
      const double t = (a.t_batched ? a.t[row] : a.t[j]) / (a.t_div > 0.0 ? a.t_div : 1.0);
      const double Tt = a.scale * t;
      const int k = col < a.S ? col : col - a.S;
      double gamma, im;
      if (a.node_re != nullptr) {  // fixed Talbot / Stehfest nodes
        gamma = a.node_re[k] / t;
        im = a.node_im[k] / t;
      } else {
        gamma = a.alpha - a.log_tol / (a.scale * Tt);
        im = kPi * (double)k / Tt;
      }
      if (col < a.S) {
        v = atan2(im, gamma);
      } else {
        const double a2 = gamma * gamma + im * im;
        v = asin((a2 - 1.0) / (a2 + 1.0));
      }
    }
    a.out[idx] = v;
  }
}

hipError_t launch_rep_inputs(const RepInArgs& a, hipStream_t s) {
  const int64_t total = a.B * a.Tt * (2 * a.S + a.P);
  if (total <= 0) return hipSuccess;
  const int64_t want = (total + 255) / 256;
  const unsigned grid = (unsigned)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(rep_inputs_kernel, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ linear closed-form algorithms (fixed Talbot, Stehfest)
// One thread per (point, dim) row walks its S (theta, phi) pairs; a wavefront's 64 rows are 64 * S contiguous doubles,
// so the lines it touches are shared by neighbouring threads' later iterations (L1 / L2 hits).  Not a tuned stream like
// the Fourier kernel: these two algorithms exist for coverage of the reference's nl_ilt_algorithm knob.
__global__ __launch_bounds__(256) void ilt_linear_kernel(const IltLinArgs a) {
  const int64_t rows = a.N * a.d;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x) {
    const double* th = a.theta + row * a.S;
    const double* ph = a.phi + row * a.S;
    double acc = 0.0;
    for (int k = 0; k < a.S; ++k) {
      const double rad = m::tan_0_halfpi(ph[k] / 2.0 + kPi / 4.0);
      double sn, cs;
      m::sincos_bounded(th[k], &sn, &cs);
      acc += a.wr[k] * (rad * cs) - a.wi[k] * (rad * sn);
    }
    a.x[row] = acc / a.t[row / a.d];
  }
}
hipError_t launch_ilt_linear(const IltLinArgs& a, hipStream_t s) {
  const int64_t rows = a.N * a.d;
  if (rows <= 0) return hipSuccess;
  const int64_t want = (rows + 255) / 256;
  hipLaunchKernelGGL(ilt_linear_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, s, a);
  return hipGetLastError();
}

// Planner path (round 3): F_k = (re, im) arrives SLOT-major (8*nt3, N) from the representation kernel's MFMA epilogue,
// as for the de Hoog planner (kernels_dehoog.hip, FMODE 2): a wavefront owns 64 consecutive samples of ONE dim, term k of
// its rows is one full 512-B line per array, and the kernel reads exactly d * S * 16 B per sample.  Same summation order
// as ilt_linear_kernel.  t is one device scalar (the planner's constant prediction time).
__global__ __launch_bounds__(64) void ilt_linear_slot_kernel(const IltLinSlotArgs a) {
  const int lane = threadIdx.x;
  const int64_t nsb = (a.N + 63) / 64;
  const int64_t nblk = nsb * a.d;
  const double t = a.t[0];
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int cdim = (int)(blk % a.d);
    const int64_t n = (blk / a.d) * 64 + lane;
    if (n >= a.N) continue;
    const int* ei = a.eidx + cdim * a.S;
    double acc = 0.0;
#pragma unroll 4
    for (int k = 0; k < a.S; ++k) {
      const int64_t at = (int64_t)ei[k] * a.N + n;
      acc += a.wr[k] * a.fre[at] - a.wi[k] * a.fim[at];
    }
    a.x[n * a.d + cdim] = acc / t;
  }
}
hipError_t launch_ilt_linear_slot(const IltLinSlotArgs& a, hipStream_t s) {
  if (a.N <= 0 || a.d <= 0) return hipSuccess;
  if (!a.fre || !a.fim || !a.eidx || !a.wr || !a.wi || !a.t || !a.x) return hipErrorInvalidValue;
  const int64_t want = (a.N + 63) / 64 * a.d;
  hipLaunchKernelGGL(ilt_linear_slot_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(64), 0, s, a);
  return hipGetLastError();
}

// backward: x = (1/t) sum_k (wr_k R_k cos(theta_k) - wi_k R_k sin(theta_k)),  R = tan(phi/2 + pi/4),  R' = (1 + R^2) / 2:
//   d x / d theta_k = -(1/t) R (wr sin + wi cos),    d x / d phi_k = (1/t) (wr cos - wi sin) (1 + R^2) / 2
// One thread per (row, term) element: reads theta, phi, writes both gradients, fully coalesced (the reference trains the
// representation function through whichever ilt_algorithm is configured, train_utils.py:388-407).
__global__ __launch_bounds__(256) void ilt_linear_bwd_kernel(const IltLinBwdArgs a) {
  const int64_t total = a.N * a.d * a.S;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / a.S;
    const int k = (int)(e - row * a.S);
    const double rad = m::tan_0_halfpi(a.phi[e] / 2.0 + kPi / 4.0);
    double sn, cs;
    m::sincos_bounded(a.theta[e], &sn, &cs);
    const double g = a.gx[row] / a.t[row / a.d];
    a.gtheta[e] = -g * rad * (a.wr[k] * sn + a.wi[k] * cs);
    a.gphi[e] = g * (a.wr[k] * cs - a.wi[k] * sn) * (0.5 * (1.0 + rad * rad));
  }
}
hipError_t launch_ilt_linear_bwd(const IltLinBwdArgs& a, hipStream_t s) {
  const int64_t total = a.N * a.d * a.S;
  if (total <= 0) return hipSuccess;
  const int64_t want = (total + 255) / 256;
  hipLaunchKernelGGL(ilt_linear_bwd_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ Fourier series
// x[n,c] = e^{gamma t}/T * sum_k w_k Re(F_k e^{i pi k/scale}),  w_0 = 1/2, t/T = 1/scale for every t.
// Rows (n,c) are contiguous runs of S doubles in theta/phi.  A block streams ROWS rows with perfectly
// coalesced loads (lane i <-> flat element i), parks the per-element contribution in LDS and lets one
// thread per row add its S terms (row stride padded odd: conflict-free ds_read_b64).

// Round 3, measured and not kept: a LINE-ALIGNED form (tiles of 256 rows = S passes of exactly 256 consecutive doubles, so
// every wavefront load is four whole 128-B lines instead of straddling a fifth as the 2040-B passes of S = 17 do; the lane's
// term index then advances per pass and its phase constants are rebuilt from k with five integer instructions; 117 VGPRs,
// no scratch, wait counts 14-18): 0.192-0.209 ms against 0.188-0.194 ms for this kernel on the same box, and this kernel
// at S = 16 (2048-B passes) moves its bytes only 6 % faster than at S = 17 -- line straddling is not what holds the
// stream at 84 % of the bare read rate.
// DBG: 0 = product; 1 / 2 = timing experiments (memory only / arithmetic only), instantiated by the tools build only
// (-DNLC_ILT_EXPERIMENTS=1)
// ITERS > 0: passes per tile known at compile time -> the pass loop is fully unrolled (straight-line code is
// what lets the compiler keep counted s_waitcnt vmcnt(N) instead of draining the pipeline); 0: runtime loop.
//
// Whole tiles run as ONE continuous load pipeline per block: the refills issued in the last UB passes of a tile are
// the first UB passes of the block's NEXT tile, so 2*UB loads per lane stay in flight through the tile's barriers
// and its row-sum phase (first version: the pipeline drained at every tile, 16 passes; the kernel ran at
// memory-only time + arithmetic time, 0.168 + 0.073 ms at N = 655 360).  The row's t is loaded at the top of the tile
// (older than the tile's refills, so the row-sum waits with a counted vmcnt, not vmcnt(0)); row / d is a wave-uniform 64-bit
// division plus a 32-bit one per thread, and the per-row scale e^{gamma t}/T uses the refined reciprocal.
// split in two so the pipeline can consume a slot (ilt_args) BEFORE refilling it and finish the arithmetic after:
// the refill then lands in the registers it just freed (one register set for the 2*UB loads in flight, not two)
__device__ __forceinline__ void ilt_args(double t_u, double p_u, double psi, double* xt, double* xp) {
  *xt = t_u + psi;
  *xp = p_u / 2.0 + kPi / 4.0;
}
// per-lane constants of the Fourier phase e^{i pi k t/T}: with scale = 2 it is i^k, i.e. cos(theta + k pi/2) =
// +-cos(theta + (k & 1) pi/2): the odd shift rides in the range reduction, the sign in the term weight
struct IltLane {
  double psi;      // added to theta (0 when scale == 2)
  double half_m;   // m/2
  double dm;       // m = quarter-turn parity
  double wk;       // +-1/2 for k = 0, +-1 otherwise
};
__device__ __forceinline__ IltLane ilt_lane(int k, double scale) {
  IltLane L;
  const bool pow_i = scale == 2.0;
  const double wk = k == 0 ? 0.5 : 1.0;
  if (pow_i) {
    L.psi = 0.0;
    L.dm = (double)(k & 1);
    L.wk = (k & 2) ? -wk : wk;
  } else {
    double psi = kPi * (double)k / scale;
    psi -= 2.0 * kPi * rint(psi / (2.0 * kPi));
    L.psi = psi;
    L.dm = 0.0;
    L.wk = wk;
  }
  L.half_m = 0.5 * L.dm;
  return L;
}
// fixed Talbot / Stehfest on the same stream (round 3): w_re Re F - w_im Im F = |w| R cos(theta + arg w), so a linear
// algorithm is the general-phase path with psi_k = arg w_k, weight |w_k| (a real weight keeps its sign and psi = 0) and the
// row scale 1/t
__device__ __forceinline__ IltLane ilt_lane_linear(double wr, double wi) {
  IltLane L;
  L.psi = wi == 0.0 ? 0.0 : atan2(wi, wr);
  L.wk = wi == 0.0 ? wr : hypot(wr, wi);
  L.dm = 0.0;
  L.half_m = 0.0;
  return L;
}
// the trig constants, each pinned in an SGPR pair for the whole kernel
__device__ __forceinline__ m::IltTrigK ilt_trig_k_sgpr() {
  m::IltTrigK K = m::ilt_trig_k();
#define NLC_PIN(x) asm volatile("" : "+s"(x))
  // (the leading coefficient of each polynomial meets a second constant in its first Horner step, and an FP64 VALU
  // instruction reads at most one scalar operand: those three stay in VGPRs)
#define NLC_PINV(x) asm volatile("" : "+v"(x))
  NLC_PINV(K.s[0]);
  NLC_PINV(K.c4[0]);
  NLC_PINV(K.c2[0]);
#undef NLC_PINV
#pragma unroll
  for (int i = 1; i < 6; ++i) NLC_PIN(K.s[i]);
#pragma unroll
  for (int i = 1; i < 6; ++i) NLC_PIN(K.c4[i]);
#pragma unroll
  for (int i = 1; i < 8; ++i) NLC_PIN(K.c2[i]);
  NLC_PIN(K.pio4_hi);
  NLC_PIN(K.pio4_lo);
  NLC_PIN(K.pio2_hi);
  NLC_PIN(K.pio2_lo);
  NLC_PIN(K.inv_pi);
  NLC_PIN(K.round_shift);
  // den_min stays a visible literal: an opaque value would have to be canonicalised before every v_max_f64
#undef NLC_PIN
  return K;
}
__device__ __forceinline__ double ilt_term2(const m::IltTrigK& K, double xt, double xp, const IltLane& L) {
  double num, den;
  m::tan_parts_short(K, xp, &num, &den);
  const double cs = m::cos_plus_mpio2(K, xt, L.half_m, L.dm);
  return (L.wk * num) * cs * m::rcp_refined(den);
}
__device__ __forceinline__ double ilt_term(const m::IltTrigK& K, double t_u, double p_u, const IltLane& L) {
  double xt, xp;
  ilt_args(t_u, p_u, L.psi, &xt, &xp);
  return ilt_term2(K, xt, xp, L);
}
__device__ __forceinline__ double ilt_row_scale(const IltArgs& a, double t) {
  // once per row, but at one row per 17 terms its IEEE divisions and libm exp were ~10 % of the kernel's instructions
  const double Tt = a.scale * t;
  const double gamma = a.alpha - m::div_fast(a.log_tol, a.scale * Tt);
  return m::div_fast(m::exp_d(gamma * t), Tt);
}
// LDS-only workgroup barrier: ds traffic drained, global loads left in flight (a __syncthreads() would also be a
// global-memory fence and wait for vmcnt(0), i.e. drain the prefetch pipeline)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 4 waves per SIMD (<= 128 VGPRs) where the unrolled tile fits without spilling, 3 otherwise
template <int DBG, int ITERS, bool LIN = false>
__global__ __launch_bounds__(256, (ITERS == 8 || ITERS == 16) ? 4 : 3) void ilt_fourier_kernel(const IltArgs a) {
  extern __shared__ double val[];  // [rows][SP]
  const int S = a.S;
  const int SP = S | 1;
  // A pass covers RPP = 256/S whole rows = RPP*S consecutive doubles, one per active thread, so a thread's
  // term index k never changes (phase/weight stay in registers) and its row advances by RPP per pass.
  const int rpp = a.rpp, iters = ITERS > 0 ? ITERS : a.iters, rows = rpp * iters;
  const int act = rpp * S;
  const bool active = (int)threadIdx.x < act;
  const int k = (int)threadIdx.x % S, rloc = (int)threadIdx.x / S;
  const IltLane L = LIN ? ilt_lane_linear(a.lin_wr[k], a.lin_wi[k]) : ilt_lane(k, a.scale);
  const m::IltTrigK K = ilt_trig_k_sgpr();
  const double psi = L.psi;
  const int64_t rows_total = a.N * a.d;
  const int64_t nblk = (rows_total + rows - 1) / rows;
  constexpr int UB = 8;  // pipeline depth in passes; the launcher makes iters a multiple of UB
  // (LIN: t_div is the model's time normalisation when the model forward drives this kernel, 1 otherwise)
  auto row_scale = [&](double t) { return LIN ? m::div_fast(a.t_div, t) : ilt_row_scale(a, t); };
  int64_t blk = blockIdx.x;

  if constexpr (ITERS > 0) {
    const int64_t nfull = rows_total / rows;  // whole tiles
    // No divergent region may contain the pipeline's loads: a skipped-region path makes the compiler's wait-count
    // bookkeeping fall back to vmcnt(0) at the join, which drains the pipeline.  So EVERY thread runs the pass loop
    // (the block's idle tail threads alias the last active one and only skip the LDS write), and every thread
    // loads a t (clamped to the tile's last row).
    const int tid_ld = active ? (int)threadIdx.x : act - 1;
    const int tid_row = (int)threadIdx.x < rows ? (int)threadIdx.x : rows - 1;
    double th[UB], ph[UB];
    const double* __restrict__ tp = a.theta + tid_ld;
    const double* __restrict__ pp = a.phi + tid_ld;
    if (blk < nfull) {
      tp += blk * rows * S;
      pp += blk * rows * S;
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (DBG == 2) {
          th[u] = ph[u] = a.alpha * (double)threadIdx.x;
          continue;
        }
        th[u] = __builtin_nontemporal_load(tp + (int64_t)act * u);
        ph[u] = __builtin_nontemporal_load(pp + (int64_t)act * u);
        // issue order = the loop's refill order, so the wait counts at the loop head agree on both entries
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (; blk < nfull; blk += gridDim.x) {
      const int64_t row0 = blk * rows;
      // this thread's output row: t first (oldest load of the tile)
      const int64_t n0 = row0 / a.d;  // wave-uniform
      const unsigned xr = (unsigned)(row0 - n0 * a.d) + (unsigned)tid_row;
      const double t_row = a.t[n0 + xr / (unsigned)a.d];  // 32-bit division, once per tile
      // successor tile of this block; without one the refills (values unused) all aim at ONE element of the
      // input -- re-reading this tile instead cost 32 KB of HBM reads per block, +9 % traffic (PMC FETCH_SIZE)
      const int64_t nxt = blk + gridDim.x;
      const bool has_next = nxt < nfull;
      const int64_t actn = has_next ? act : 0;
      const double* __restrict__ tn = has_next ? tp + (int64_t)gridDim.x * rows * S : a.theta;  // element 0: always valid
      const double* __restrict__ pn = has_next ? pp + (int64_t)gridDim.x * rows * S : a.phi;
#pragma unroll
      for (int i = 0; i < ITERS; ++i) {
        const int u = i % UB;
        double xt, xp;
        if (DBG == 1) {
          xt = th[u];
          xp = ph[u];
        } else {
          ilt_args(th[u], ph[u], psi, &xt, &xp);
        }
        asm volatile("" : "+v"(xt), "+v"(xp));  // slot u is consumed here ...
        __builtin_amdgcn_sched_barrier(0);
        if (DBG == 2) {  // timing experiment: arithmetic only (run-time values, nothing to fold)
          th[u] = a.alpha * (double)(threadIdx.x + i) + (double)blk * 1e-7;
          ph[u] = a.alpha * (double)(threadIdx.x + 3 * i);
        } else if (i + UB < ITERS) {            // ... and refilled into the same registers
          th[u] = __builtin_nontemporal_load(tp + (int64_t)act * (i + UB));
          ph[u] = __builtin_nontemporal_load(pp + (int64_t)act * (i + UB));
        } else {
          th[u] = __builtin_nontemporal_load(tn + actn * (i + UB - ITERS));
          ph[u] = __builtin_nontemporal_load(pn + actn * (i + UB - ITERS));
        }
        __builtin_amdgcn_sched_barrier(0);
        const double v = DBG == 1 ? xt + xp : ilt_term2(K, xt, xp, L);
        if (active) val[(rloc + rpp * i) * SP + k] = v;
        __builtin_amdgcn_sched_barrier(0);  // keep the refill of slot u next to its use: no load clustering
      }
      tp = tn;
      pp = pn;
      lds_barrier();
      if ((int)threadIdx.x < rows) {
        const double* v = val + threadIdx.x * SP;
        double acc = 0.0;
        for (int kk = 0; kk < S; ++kk) acc += v[kk];
        a.x[row0 + threadIdx.x] = row_scale(t_row) * acc;
      }
      lds_barrier();
    }
    // the ragged last tile (if any) falls through to the generic loop of the block whose turn it is
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  for (; blk < nblk; blk += gridDim.x) {
    const int64_t row0 = blk * rows;
    const int rows_here = (int)((rows_total - row0 < rows) ? (rows_total - row0) : rows);
    const double* __restrict__ tp = a.theta + row0 * S + threadIdx.x;
    const double* __restrict__ pp = a.phi + row0 * S + threadIdx.x;
    // Software pipeline of depth UB over the passes: pass i is computed while the loads of passes
    // i+1 .. i+UB are in flight (2*UB 8-byte loads per lane outstanding at all times).
    // Per element: Re(F_k e^{i pi k/scale}) = tan(phi/2 + pi/4) cos(theta + pi k/scale); tan as
    // (cos a + sin a)/(cos a - sin a) (no range reduction, no branch), its division folded into the product.
    if (active && rloc < rows_here) {
      double th[UB], ph[UB];
      // No branch may sit between a load and its use, or the compiler falls back to s_waitcnt vmcnt(0) and the
      // pipeline collapses: out-of-range passes re-load the last valid pass (cache hit) instead of being skipped.
      const int last_i = (rows_here - 1 - rloc) / rpp;
      auto fetch = [&](int u, int i) {
        const int ic = i < last_i ? i : last_i;
        th[u] = __builtin_nontemporal_load(tp + (int64_t)act * ic);
        ph[u] = __builtin_nontemporal_load(pp + (int64_t)act * ic);
      };
#pragma unroll
      for (int u = 0; u < UB; ++u) fetch(u, u);
      for (int i0 = 0; i0 < iters; i0 += UB) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int i = i0 + u;
          const int r = rloc + rpp * i;
          const double t_u = th[u], p_u = ph[u];
          fetch(u, i + UB);
          const double v = DBG == 1 ? t_u + p_u : ilt_term(K, t_u, p_u, L);
          if (r < rows_here) val[r * SP + k] = v;
        }
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < rows_here) {
      const double* v = val + threadIdx.x * SP;
      double acc = 0.0;
      for (int kk = 0; kk < S; ++kk) acc += v[kk];
      const int64_t row = row0 + threadIdx.x;
      a.x[row] = row_scale(a.t[row / a.d]) * acc;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ Fourier series, one LANE per row (round 6)
// The kernel above gives a lane ONE term of a row and pays for it: a value parked in LDS per term, two workgroup barriers and a
// row-sum phase per tile, 64-bit address arithmetic per load, per-lane phase constants -- 57 + 5 FP64 / integer instructions per
// 16-byte (theta, phi) pair, and its arithmetic-only time (0.17 ms at N = 655 360) sits ABOVE its memory-only time (0.154 ms).
// Here a lane owns a whole ROW (one (point, dim): S terms) and a wavefront a tile of 64 consecutive rows = 64 S consecutive doubles
// of theta and of phi.  The tile lands in the wavefront's private LDS region by DIRECT global -> LDS loads
// (global_load_lds_dwordx4: 16 B per lane, 1 KB per instruction, no VGPR, no address arithmetic beyond the tile base), is read
// back row-wise (stride S doubles, S odd: conflict-free ds_read_b64) and summed in a register: no barrier of any kind (a
// wavefront only waits for its own loads), no LDS writes by the ALU, the term index -- so the quarter turn i^k of the phase and
// the sign of the weight -- a compile-time constant, the tangent as one rational (m::tan_parts_rat): ~38 instructions per pair.
// Eight wavefronts per CU (2 x 34 KB regions per workgroup of four at S = 17), each with a whole tile in flight while its SIMD
// partner computes.  Instances: S = 17 and S = 33 (the reference's default and its de Hoog ablation's term count) with scale = 2
// (torchlaplace's default: e^{i pi k t / T} = i^k); anything else -- other term counts, another scale, the linear algorithms,
// unaligned inputs -- keeps ilt_fourier_kernel.
// GEN = false: scale == 2, the phase of term k is i^k (compile-time quarter turn and sign, w_0 = 1/2).
// GEN = true: a per-term phase psi_k and weight w_k from a small table in LDS (`tab`: psi_0, w_0, psi_1, w_1, ...; every lane
// reads the same address: a broadcast) -- the Fourier series at another scale (psi_k = pi k / scale) and the linear algorithms
// (fixed Talbot / Stehfest: w_re Re F - w_im Im F = |w| R cos(theta + arg w), kernels above).
template <int S, bool GEN, class LD>
__device__ __forceinline__ double ilt_row_sum(const m::IltRowK& K, const double* tab, LD ld) {
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < S; ++k) {
    double th, ph;
    ld(k, &th, &ph);
    // tan(phi/2 + pi/4): the argument rounded as the reference rounds it (w_nl.py: tan of the SUM), then a = x - pi/4
    const double x = fma(ph, 0.5, kPi / 4.0);
    const double a = (x - K.pio4_hi) - K.pio4_lo;
    double num, den;
    m::tan_parts_rat(K, a, &num, &den);
    const double r = m::rcp_refined(den);
    if (GEN) {
      const double cs = m::cos_or_sin_reduced<0>(K, th + tab[2 * k]);
      acc = fma((tab[2 * k + 1] * num) * cs, r, acc);
    } else {
      // Re(F_k i^k) = |F_k| cos(theta + k pi/2): k & 1 picks the polynomial, ((k + 1) >> 1) & 1 the sign; w_0 = 1/2
      const double cs = (k & 1) ? m::cos_or_sin_reduced<1>(K, th) : m::cos_or_sin_reduced<0>(K, th);
      const bool neg = (((k + 1) >> 1) & 1) != 0;
      const double nc = (k == 0 ? 0.5 * num : num) * cs;
      acc = fma(neg ? -nc : nc, r, acc);
    }
  }
  return acc;
}
__device__ __forceinline__ m::IltRowK ilt_row_k_sgpr() {
  m::IltRowK K = m::ilt_row_k();
#define NLC_PIN(x) asm volatile("" : "+s"(x))
#define NLC_PINV(x) asm volatile("" : "+v"(x))
  // (an FP64 VALU instruction reads at most one scalar operand: the leading coefficient of a Horner chain meets a second
  // constant in its first step and stays in a VGPR)
  NLC_PINV(K.pn[0]);
  NLC_PINV(K.c2[0]);
  NLC_PINV(K.s2[0]);
#pragma unroll
  for (int i = 0; i < 4; ++i) NLC_PIN(K.qn[i]);
#pragma unroll
  for (int i = 1; i < 3; ++i) NLC_PIN(K.pn[i]);
#pragma unroll
  for (int i = 1; i < 8; ++i) NLC_PIN(K.c2[i]);
#pragma unroll
  for (int i = 1; i < 8; ++i) NLC_PIN(K.s2[i]);
  NLC_PIN(K.pio4_hi);
  NLC_PIN(K.pio4_lo);
  NLC_PIN(K.pi_hi);
  NLC_PIN(K.pi_lo);
  NLC_PIN(K.inv_pi);
  NLC_PIN(K.round_shift);
#undef NLC_PIN
#undef NLC_PINV
  return K;
}

// The tile loads and the counted waits are written out: the loop keeps BOTH arrays' next tiles in flight while it computes
// (theta of tile n + 1 is requested the moment theta of tile n has been read out of LDS, phi likewise), which needs
// s_waitcnt vmcnt(N) with N = the loads issued behind the awaited ones -- the compiler's own bookkeeping answers an LDS read
// behind an LDS-DMA load with vmcnt(0).  Nothing else in the loop is a vector-memory load the compiler counts (the row's t
// rides in the same hand-counted queue); the store of x needs no wait.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void ilt_lds_load16(const void* g, unsigned lds_addr) {  // 64 lanes x 16 B -> LDS [lds_addr, + 1 KB)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(g), "s"(lds_addr) : "memory", "m0");
}
#pragma clang diagnostic pop
template <int TILE>
__device__ __forceinline__ void ilt_lds_load_tile(const char* g_lane, unsigned lds_addr, int lane) {
  constexpr int NLD = TILE / 1024, REM = TILE % 1024;  // whole-wave 16-byte loads + the lanes of the last one
#pragma unroll
  for (int i = 0; i < NLD; ++i) ilt_lds_load16(g_lane + i * 1024, lds_addr + i * 1024);
  if (REM != 0) {
    // (every lane issues: the lanes past the tile re-read its last 16 bytes into the pad behind it -- a masked load would sit in
    // a divergent region and change the number of outstanding loads per lane group)
    const char* g = (lane < REM / 16) ? g_lane + NLD * 1024 : g_lane + NLD * 1024 - (lane - (REM / 16 - 1)) * 16;
    ilt_lds_load16(g, lds_addr + NLD * 1024);
  }
}

// DBG (tools build only): 1 = loads + the plain sum of what they brought (memory only), 2 = arithmetic on run-time values, no loads
// DEPTH: (theta, phi) region pairs per wavefront = tiles it keeps in flight.  1: eight wavefronts per CU, each requesting the next
// tile's arrays as it reads the current ones out; 2 (S <= 17): four wavefronts per CU with two region pairs each -- a landed tile
// no longer waits for its wavefront to finish the tile before it, at half the FP64 issue rate (one wavefront per SIMD), which
// this kernel can afford (arithmetic-only 0.043 ms of 0.16).
template <int S, bool GEN, int DBG, int DEPTH>
__global__ __launch_bounds__(256, (S <= 17 && DEPTH == 1) ? 2 : 1) void ilt_fourier_rows_kernel(const IltArgs a) {
  static_assert(S % 2 == 1, "row stride S doubles must be odd: conflict-free row-wise reads, 16-byte tile sizes");
  extern __shared__ __attribute__((aligned(16))) char rows_lds[];
  constexpr int TILE = 64 * S * 8;                          // bytes of one array's tile
  constexpr int SLOT = (TILE + 1023) / 1024 * 1024;        // its LDS slot: the last (half) load writes a full KB
  constexpr int LPT = SLOT / 1024;                          // loads per tile and array
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (wave-uniform: scalar loop, scalar M0)
  char* lt0 = rows_lds + wave * (DEPTH * 2 * SLOT);
  const unsigned lt_addr0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lt0;
  const m::IltRowK K = ilt_row_k_sgpr();
  const int64_t rows_total = a.N * a.d;
  const int64_t nfull = rows_total / 64;
  const int64_t W = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + wave;
  // GEN: the per-term (phase, weight) table behind the four wavefronts' slots -- the launch's only barrier
  const double* tab = (const double*)(rows_lds + 4 * (DEPTH * 2 * SLOT));
  const bool lin = a.lin_wr != nullptr;
  if (GEN) {
    if ((int)threadIdx.x < S) {
      const int k = threadIdx.x;
      const IltLane L = lin ? ilt_lane_linear(a.lin_wr[k], a.lin_wi[k]) : ilt_lane(k, a.scale);
      // (ilt_lane at scale == 2 describes the quarter turns by dm / a signed weight: spell them out as a phase)
      double* tw = (double*)(rows_lds + 4 * (DEPTH * 2 * SLOT));
      tw[2 * k] = L.psi + L.dm * (kPi / 2.0);
      tw[2 * k + 1] = L.wk;
    }
    __syncthreads();
  }
  // the row scale: e^{gamma t} / T of the Fourier series, 1 / t (times the model's time normalisation) of the linear algorithms
  auto row_scale = [&](double t) { return (GEN && lin) ? m::div_fast(a.t_div, t) : ilt_row_scale(a, t); };
  // the row's point index n = row / d, kept incrementally: row advances by 64 W per tile
  const unsigned d = (unsigned)a.d;
  const int64_t step = 64 * W;
  const int64_t step_n = step / d;
  const unsigned step_r = (unsigned)(step - step_n * d);
  int64_t row = w0 * 64 + lane;
  int64_t n = row / d;
  unsigned rem = (unsigned)(row - n * d);
  double t_prev = __builtin_nan(""), sc = 0.0;  // the row scale e^{gamma t}/T is rebuilt only when the row's t changes

  if (DBG == 2) {
    for (int64_t tile = w0; tile < nfull; tile += W) {
      const double acc = ilt_row_sum<S, GEN>(K, tab, [&](int k, double* th, double* ph) {
        *th = a.alpha * (double)(lane + k) + (double)tile * 1e-7;
        *ph = a.alpha * (double)(lane + 3 * k);
      });
      __builtin_nontemporal_store(row_scale(0.125) * acc, a.x + row);
      row += step;
    }
  } else if (w0 < nfull) {
    // queue of this wavefront's vector-memory loads, oldest first, at the head of iteration n:
    //   theta_n (LPT)  phi_n (LPT)            [+ the store of x_(n-1), which no wait below depends on]
    // The row's t is requested at the head of the iteration that uses it and consumed behind a wait of its own: its register must
    // not travel through a loop-carried copy -- the compiler does not know the load is asynchronous and would copy a register
    // the data has not reached yet.
    auto req_theta = [&](int64_t tile, bool real, int par) {
      const char* g = real ? (const char*)(a.theta + tile * (64 * S)) + lane * 16 : (const char*)a.theta;  // no successor: the array's head
      ilt_lds_load_tile<TILE>(g, lt_addr0 + par * (2 * SLOT), real ? lane : 0);
    };
    auto req_phi = [&](int64_t tile, bool real, int par) {
      const char* g = real ? (const char*)(a.phi + tile * (64 * S)) + lane * 16 : (const char*)a.phi;
      ilt_lds_load_tile<TILE>(g, lt_addr0 + par * (2 * SLOT) + SLOT, real ? lane : 0);
    };
    auto req_t = [&](int64_t nn) {
      double v;
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(a.t + nn) : "memory");
      return v;
    };
    req_theta(w0, true, 0);
    req_phi(w0, true, 0);
    if (DEPTH == 2) {
      req_theta(w0 + W, w0 + W < nfull, 1);
      req_phi(w0 + W, w0 + W < nfull, 1);
    }
    int par = 0;  // the region pair of the current tile (DEPTH = 2: alternating)
    for (int64_t tile = w0; tile < nfull; tile += W) {
      double t_row = req_t(n);                        // queue: theta_n, phi_n, [theta_(n+1), phi_(n+1),] t_n
      const int64_t nxt = tile + DEPTH * W;
      const bool has_next = nxt < nfull;
      const double* lrow_t = (const double*)(lt0 + par * (2 * SLOT)) + lane * S;
      const double* lrow_p = (const double*)(lt0 + par * (2 * SLOT) + SLOT) + lane * S;
      // the next tile's point index
      int64_t n_next = n + step_n;
      unsigned rem_next = rem + step_r;
      if (rem_next >= d) {
        rem_next -= d;
        n_next += 1;
      }
      // ---- theta_n has landed when at most the loads behind it are outstanding: phi_n, [the next tile's two arrays,] t_n
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((2 * DEPTH - 1) * LPT + 1) : "memory");
      double th[S], ph[S];
#pragma unroll
      for (int k = 0; k < S; ++k) th[k] = lrow_t[k];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      req_theta(nxt, has_next, par);                  // queue: phi_n, [..,] t_n, theta_(n+DEPTH)
      // ---- phi_n has landed when at most [the next tile's two arrays,] t_n and theta_(n+DEPTH) are outstanding
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((2 * DEPTH - 1) * LPT + 1) : "memory");
#pragma unroll
      for (int k = 0; k < S; ++k) ph[k] = lrow_p[k];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      req_phi(nxt, has_next, par);                    // queue: [..,] t_n, theta_(n+DEPTH), phi_(n+DEPTH)
      double acc;
      if (DBG == 1) {
        acc = 0.0;
#pragma unroll
        for (int k = 0; k < S; ++k) acc += th[k] + ph[k];
      } else {
        acc = ilt_row_sum<S, GEN>(K, tab, [&](int k, double* t_o, double* p_o) {
          *t_o = th[k];
          *p_o = ph[k];
        });
      }
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(t_row) : "n"(2 * LPT) : "memory");  // t_n: the next tile's loads are behind it
      if (t_row != t_prev) {  // (planning and training batches share one t: skipped after the first tile)
        sc = row_scale(t_row);
        t_prev = t_row;
      }
      __builtin_nontemporal_store(sc * acc, a.x + row);
      row += step;
      n = n_next;
      rem = rem_next;
      if (DEPTH == 2) par ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last iteration's stand-in loads
  }
  // the ragged last tile (fewer than 64 rows): the wavefront whose turn it would be reads its rows straight from memory
  if (DBG != 2 && nfull * 64 < rows_total && nfull % W == w0) {
    const int64_t r = nfull * 64 + lane;
    if (r < rows_total) {
      const double* rt = a.theta + r * S;
      const double* rp = a.phi + r * S;
      const double acc = ilt_row_sum<S, GEN>(K, tab, [&](int k, double* th, double* ph) {
        *th = rt[k];
        *ph = rp[k];
      });
      a.x[r] = row_scale(a.t[r / a.d]) * acc;
    }
  }
}


// More than 64 KB of dynamic LDS per workgroup needs hipFuncAttributeMaxDynamicSharedMemorySize, per function AND per device:
// remembered per (kernel instance, device) so that a process that plans on several GPUs sets it on each.
template <class F>
static bool lds_attr_once(F* fn, size_t shmem, std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  const unsigned long long bit = 1ull << dev;
  if (done.load(std::memory_order_acquire) & bit) return true;
  if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess) return false;
  done.fetch_or(bit, std::memory_order_release);
  return true;
}
// true when the row-per-lane kernel takes the launch: an odd term count 3 .. 33 (row-wise LDS reads are conflict-free for an
// odd stride; the reference's default 17, its de Hoog ablation's 33, fixed Talbot's 17) and 16-byte aligned inputs
template <int S, bool GEN, int DBG, int DEPTH = 1>
static bool launch_rows_instance(const IltArgs& a, hipStream_t s, hipError_t* err) {
  constexpr int SLOT = (64 * S * 8 + 1023) / 1024 * 1024;
  constexpr size_t shmem = (size_t)4 * DEPTH * 2 * SLOT + 2 * S * 8;  // four wavefronts' theta / phi slots + the (phase, weight) table
  static std::atomic<unsigned long long> attr_done{0};
  if (!lds_attr_once(ilt_fourier_rows_kernel<S, GEN, DBG, DEPTH>, shmem, attr_done)) return false;
  const int64_t tiles = (a.N * a.d + 63) / 64;
  const int per_cu = (S <= 17 && DEPTH == 1) ? 2 : 1;  // workgroups of four wavefronts per CU (launch bounds, LDS)
  int64_t grid = (tiles + 3) / 4;
  if (grid > 256 * per_cu) grid = 256 * per_cu;
  hipLaunchKernelGGL((ilt_fourier_rows_kernel<S, GEN, DBG, DEPTH>), dim3((unsigned)grid), dim3(256), shmem, s, a);
  *err = hipGetLastError();
  return true;
}
static bool launch_ilt_fourier_rows(const IltArgs& a, hipStream_t s, hipError_t* err) {
  if (a.S < 3 || a.S > 33 || (a.S & 1) == 0) return false;
  if (a.lin_wr != nullptr && a.lin_wi == nullptr) return false;
  if ((((uintptr_t)a.theta) | ((uintptr_t)a.phi)) & 15) return false;  // 16-byte loads
  const bool gen = a.lin_wr != nullptr || a.scale != 2.0;
#if NLC_ILT_EXPERIMENTS
  static const int dbg_env = [] {
    const char* ev = std::getenv("NLC_ILT_DBG");
    return ev ? std::atoi(ev) : 0;
  }();
  static const int rows_env = [] {
    const char* ev = std::getenv("NLC_ILT_ROWS");
    return ev ? std::atoi(ev) : 1;
  }();
  if (!rows_env) return false;
  static const int depth_env = [] {
    const char* ev = std::getenv("NLC_ILT_DEPTH");
    return ev ? std::atoi(ev) : 2;
  }();
  if (a.S == 17 && !gen && depth_env == 1) {
    if (dbg_env == 1) return launch_rows_instance<17, false, 1, 1>(a, s, err);
    if (dbg_env == 2) return launch_rows_instance<17, false, 2, 1>(a, s, err);
    return launch_rows_instance<17, false, 0, 1>(a, s, err);
  }
  if (a.S == 17 && !gen && dbg_env == 1) return launch_rows_instance<17, false, 1, 2>(a, s, err);
  if (a.S == 17 && !gen && dbg_env == 2) return launch_rows_instance<17, false, 2, 2>(a, s, err);
#endif
  switch (a.S) {
  // two tiles in flight per wavefront where two region pairs fit four wavefronts' LDS (S <= 17): 0.160 -> 0.154 ms at S = 17
#define NLC_ROWS_CASE(SS)                                                                                    \
  case SS:                                                                                                   \
    return gen ? launch_rows_instance<SS, true, 0, (SS <= 17 ? 2 : 1)>(a, s, err)                            \
               : launch_rows_instance<SS, false, 0, (SS <= 17 ? 2 : 1)>(a, s, err);
    NLC_ROWS_CASE(3) NLC_ROWS_CASE(5) NLC_ROWS_CASE(7) NLC_ROWS_CASE(9) NLC_ROWS_CASE(11) NLC_ROWS_CASE(13) NLC_ROWS_CASE(15)
    NLC_ROWS_CASE(17) NLC_ROWS_CASE(19) NLC_ROWS_CASE(21) NLC_ROWS_CASE(23) NLC_ROWS_CASE(25) NLC_ROWS_CASE(27) NLC_ROWS_CASE(29)
    NLC_ROWS_CASE(31) NLC_ROWS_CASE(33)
#undef NLC_ROWS_CASE
  }
  return false;
}

hipError_t launch_ilt_fourier(const IltArgs& a_in, hipStream_t s) {
  IltArgs a = a_in;
  const int64_t rows_total = a.N * a.d;
  if (rows_total <= 0) return hipSuccess;
  if (a.S > 256) return hipErrorInvalidValue;
  {
    hipError_t e2 = hipSuccess;
    if (launch_ilt_fourier_rows(a, s, &e2)) return e2;
  }
#if NLC_ILT_EXPERIMENTS
  if (a.lin_wr != nullptr && std::getenv("NLC_ILT_LINEAR_ROWS")) return hipErrorInvalidValue;  // time the one-thread-per-row kernel
#endif
  // rows per block tile = rpp * iters: one thread per row for the final sum (<= 256), LDS tile under 60 KiB
  const int SP = a.S | 1;
  a.rpp = 256 / a.S;
  int max_rows = (60 * 1024 / 8) / SP;
  if (max_rows > 256) max_rows = 256;
  if (a.rpp > 32) a.rpp = 32;          // rpp * 8 passes must fit the 256 row-sum threads
  a.iters = max_rows / a.rpp / 8 * 8;  // multiple of the kernel's pipeline depth
  if (a.iters < 8) return hipErrorInvalidValue;
  const int rows = a.rpp * a.iters;
  const int64_t nblk = (rows_total + rows - 1) / rows;
  // persistent grid: 1024-4096 blocks measured the same within run-to-run noise (round 1)
  int64_t cap = 2048;
  a.dbg = 0;
#if NLC_ILT_EXPERIMENTS
  // tools build only (make EXTRA_kernels_ilt=-DNLC_ILT_EXPERIMENTS=1): grid size and the two timing variants of the
  // kernel (1 memory-only, 2 arithmetic-only) from the environment, read once
  static const int64_t cap_env = [] {
    const char* ev = std::getenv("NLC_ILT_GRID");
    return (ev && std::atoll(ev) > 0) ? (int64_t)std::atoll(ev) : (int64_t)2048;
  }();
  static const int dbg_env = [] {
    const char* ev = std::getenv("NLC_ILT_DBG");
    return ev ? std::atoi(ev) : 0;
  }();
  cap = cap_env;
  a.dbg = dbg_env;
#endif
  const unsigned grid = (unsigned)(nblk < cap ? nblk : cap);
  const size_t shmem = (size_t)rows * SP * sizeof(double);
#define NLC_ILT_LAUNCH(D)                                                                              \
  switch (a.iters) {                                                                                   \
    case 8: hipLaunchKernelGGL((ilt_fourier_kernel<D, 8>), dim3(grid), dim3(256), shmem, s, a); break;   \
    case 16: hipLaunchKernelGGL((ilt_fourier_kernel<D, 16>), dim3(grid), dim3(256), shmem, s, a); break; \
    case 24: hipLaunchKernelGGL((ilt_fourier_kernel<D, 24>), dim3(grid), dim3(256), shmem, s, a); break; \
    case 32: hipLaunchKernelGGL((ilt_fourier_kernel<D, 32>), dim3(grid), dim3(256), shmem, s, a); break; \
    default: hipLaunchKernelGGL((ilt_fourier_kernel<D, 0>), dim3(grid), dim3(256), shmem, s, a); break;  \
  }
  if (a.lin_wr != nullptr) {
    // fixed Talbot / Stehfest: the same stream with per-term phase and weight from the algorithm's tables
    if (a.lin_wi == nullptr) return hipErrorInvalidValue;
    switch (a.iters) {
      case 8: hipLaunchKernelGGL((ilt_fourier_kernel<0, 8, true>), dim3(grid), dim3(256), shmem, s, a); break;
      case 16: hipLaunchKernelGGL((ilt_fourier_kernel<0, 16, true>), dim3(grid), dim3(256), shmem, s, a); break;
      case 24: hipLaunchKernelGGL((ilt_fourier_kernel<0, 24, true>), dim3(grid), dim3(256), shmem, s, a); break;
      case 32: hipLaunchKernelGGL((ilt_fourier_kernel<0, 32, true>), dim3(grid), dim3(256), shmem, s, a); break;
      default: hipLaunchKernelGGL((ilt_fourier_kernel<0, 0, true>), dim3(grid), dim3(256), shmem, s, a); break;
    }
    return hipGetLastError();
  }
#if NLC_ILT_EXPERIMENTS
  if (a.dbg == 1) {
    NLC_ILT_LAUNCH(1)
  } else if (a.dbg == 2) {
    NLC_ILT_LAUNCH(2)
  } else
#endif
  {
    NLC_ILT_LAUNCH(0)
  }
#undef NLC_ILT_LAUNCH
  return hipGetLastError();
}

// ------------------------------------------------------------------ Fourier series, backward
// x[n,c] = s(t_n) sum_k w_k R(phi_k) cos(theta_k + psi_k),  R(phi) = tan(phi/2 + pi/4) = num/den,  s = e^{gamma t}/T:
//   d x / d theta_k = -s w_k R sin(theta_k + psi_k)
//   d x / d phi_k   =  s w_k cos(theta_k + psi_k) / den^2        (R' = (1 + R^2)/2 and num^2 + den^2 = 2)
// One pass over the same (row, term) stream as the forward kernel: reads theta, phi, writes grad_theta, grad_phi
// (4 d S * 8 algorithmic bytes per point), the row's upstream gradient times s staged per tile in LDS.
// No gradient flows to t (the reference's planner and training loop never differentiate the time grid).
__global__ __launch_bounds__(256) void ilt_fourier_bwd_kernel(const IltBwdArgs a) {
  extern __shared__ double gs[];  // [rows]
  const int S = a.S;
  const int rpp = a.rpp, iters = a.iters, rows = rpp * iters;
  const int act = rpp * S;
  const bool active = (int)threadIdx.x < act;
  const int k = (int)threadIdx.x % S, rloc = (int)threadIdx.x / S;
  const IltLane L = ilt_lane(k, a.scale);
  const m::IltTrigK K = ilt_trig_k_sgpr();
  const int64_t rows_total = a.N * a.d;
  const int64_t nblk = (rows_total + rows - 1) / rows;
  constexpr int UB = 8;
  IltArgs sc{};  // the row scale only reads these
  sc.alpha = a.alpha;
  sc.log_tol = a.log_tol;
  sc.scale = a.scale;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t row0 = blk * rows;
    const int rows_here = (int)((rows_total - row0 < rows) ? (rows_total - row0) : rows);
    if ((int)threadIdx.x < rows_here) {
      const int64_t row = row0 + threadIdx.x;
      gs[threadIdx.x] = a.gx[row] * ilt_row_scale(sc, a.t[row / a.d]);
    }
    __syncthreads();
    if (active && rloc < rows_here) {
      const double* __restrict__ tp = a.theta + row0 * S + threadIdx.x;
      const double* __restrict__ pp = a.phi + row0 * S + threadIdx.x;
      double* __restrict__ gt = a.gtheta + row0 * S + threadIdx.x;
      double* __restrict__ gp = a.gphi + row0 * S + threadIdx.x;
      double th[UB], ph[UB];
      const int last_i = (rows_here - 1 - rloc) / rpp;
      const double* __restrict__ td = a.theta;  // element 0 (always valid) for the passes past the end:
      const double* __restrict__ pd = a.phi;    // no branch at the load, no wasted HBM reads
      auto fetch = [&](int u, int i) {
        const bool in = i <= last_i;
        th[u] = __builtin_nontemporal_load(in ? tp + (int64_t)act * i : td);
        ph[u] = __builtin_nontemporal_load(in ? pp + (int64_t)act * i : pd);
      };
#pragma unroll
      for (int u = 0; u < UB; ++u) fetch(u, u);
      for (int i0 = 0; i0 < iters; i0 += UB) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int i = i0 + u;
          const int r = rloc + rpp * i;
          double xt, xp;
          ilt_args(th[u], ph[u], L.psi, &xt, &xp);
          fetch(u, i + UB);
          double num, den, sn, cs;
          m::tan_parts_short(K, xp, &num, &den);
          m::sincos_plus_mpio2(K, xt, L.half_m, L.dm, &sn, &cs);
          const double inv = m::rcp_refined(den);
          if (r < rows_here) {
            const double g = gs[r] * L.wk;
            __builtin_nontemporal_store(-(g * num) * inv * sn, gt + (int64_t)act * i);
            __builtin_nontemporal_store((g * cs) * inv * inv, gp + (int64_t)act * i);
          }
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ Fourier series, backward, one LANE per row (round 6)
// The forward row kernel's shape for the gradient: a wavefront's tile of 64 rows lands in its LDS region by direct global -> LDS
// loads, a lane reads its row's S (theta, phi) pairs, writes the 2 S gradients back over them (its own row: no hazard), and the
// wavefront stores the region -- 16 B per lane, whole lines -- to gtheta / gphi.  No per-element address arithmetic, no barrier,
// the quarter turn i^k of every term a compile-time constant (sin x and cos x from ONE reduction, m::sincos_reduced), the
// tangent as one rational:  F_k = w_k R c_k(theta),  R = tan(phi/2 + pi/4),  R' = (1 + R^2) / 2,
//   c_k = cos, -sin, -cos, sin and c_k' = -sin, -cos, sin, cos for k mod 4 = 0 .. 3  (scale = 2: e^{i pi k t / T} = i^k);
//   d x / d theta_k = G w_k R c_k'(theta_k),   d x / d phi_k = G w_k R' c_k(theta_k),   G = gx e^{gamma t} / T.
// Odd term counts 3 .. 33 with scale = 2 and 16-byte aligned arrays; everything else keeps ilt_fourier_bwd_kernel.
template <int S>
__global__ __launch_bounds__(256, S <= 17 ? 2 : 1) void ilt_fourier_bwd_rows_kernel(const IltBwdArgs a) {
  static_assert(S % 2 == 1, "row stride S doubles must be odd");
  extern __shared__ __attribute__((aligned(16))) char rows_lds[];
  constexpr int TILE = 64 * S * 8;
  constexpr int SLOT = (TILE + 1023) / 1024 * 1024;
  constexpr int NLD = TILE / 1024, REM = TILE % 1024;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* lt = rows_lds + wave * (2 * SLOT);
  char* lp = lt + SLOT;
  const unsigned lt_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lt;
  const unsigned lp_addr = lt_addr + SLOT;
  double* lrow_t = (double*)lt + lane * S;
  double* lrow_p = (double*)lp + lane * S;
  const m::IltRowK K = ilt_row_k_sgpr();
  const int64_t rows_total = a.N * a.d;
  const int64_t nfull = rows_total / 64;
  const int64_t W = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + wave;
  IltArgs sc_args{};  // the row scale only reads these
  sc_args.alpha = a.alpha;
  sc_args.log_tol = a.log_tol;
  sc_args.scale = a.scale;
  const unsigned d = (unsigned)a.d;
  const int64_t step = 64 * W;
  const int64_t step_n = step / d;
  const unsigned step_r = (unsigned)(step - step_n * d);
  int64_t row = w0 * 64 + lane;
  int64_t n = row / d;
  unsigned rem = (unsigned)(row - n * d);
  double t_prev = __builtin_nan(""), sc = 0.0;
  // gradients of one row from its (theta, phi) pairs; G = gx * row scale
  auto row_grads = [&](double G, auto ld, auto st) {
#pragma unroll
    for (int k = 0; k < S; ++k) {
      double th, ph;
      ld(k, &th, &ph);
      const double x = fma(ph, 0.5, kPi / 4.0);
      const double aa = (x - K.pio4_hi) - K.pio4_lo;
      double num, den;
      m::tan_parts_rat(K, aa, &num, &den);
      const double R = num * m::rcp_refined(den);
      const double Rp = 0.5 * fma(R, R, 1.0);
      double sn, cs;
      m::sincos_reduced(K, th, &sn, &cs);
      const double Gw = k == 0 ? 0.5 * G : G;
      // k mod 4:          0      1      2      3
      const double c = (k & 1) ? sn : cs;        //  cos   -sin   -cos    sin
      const double cd = (k & 1) ? cs : sn;       // -sin   -cos    sin    cos
      const bool neg_c = (((k + 1) >> 1) & 1) != 0;
      const bool neg_cd = ((k >> 1) & 1) == 0;
      const double gt = (Gw * R) * cd, gp = (Gw * Rp) * c;
      st(k, neg_cd ? -gt : gt, neg_c ? -gp : gp);
    }
  };
  for (int64_t tile = w0; tile < nfull; tile += W) {
    ilt_lds_load_tile<TILE>((const char*)(a.theta + tile * (64 * S)) + lane * 16, lt_addr, lane);
    ilt_lds_load_tile<TILE>((const char*)(a.phi + tile * (64 * S)) + lane * 16, lp_addr, lane);
    double t_row, gx_row;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(t_row) : "v"(a.t + n) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(gx_row) : "v"(a.gx + row) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(t_row), "+v"(gx_row)::"memory");
    if (t_row != t_prev) {
      sc = ilt_row_scale(sc_args, t_row);
      t_prev = t_row;
    }
    row_grads(
        gx_row * sc,
        [&](int k, double* th, double* ph) {
          *th = lrow_t[k];
          *ph = lrow_p[k];
        },
        [&](int k, double gt, double gp) {
          lrow_t[k] = gt;
          lrow_p[k] = gp;
        });
    // the region leaves as it came: 16 B per lane, 1 KB per instruction (LDS operations of one wavefront execute in order: the
    // reads below see every lane's writes above)
    char* gtp = (char*)(a.gtheta + tile * (64 * S)) + lane * 16;
    char* gpp = (char*)(a.gphi + tile * (64 * S)) + lane * 16;
    typedef double v2dd __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const v2dd vt = *(const v2dd*)(lt + i * 1024 + lane * 16);
      const v2dd vp = *(const v2dd*)(lp + i * 1024 + lane * 16);
      __builtin_nontemporal_store(vt, (v2dd*)(gtp + i * 1024));
      __builtin_nontemporal_store(vp, (v2dd*)(gpp + i * 1024));
    }
    if (REM != 0 && lane < REM / 16) {
      const v2dd vt = *(const v2dd*)(lt + NLD * 1024 + lane * 16);
      const v2dd vp = *(const v2dd*)(lp + NLD * 1024 + lane * 16);
      __builtin_nontemporal_store(vt, (v2dd*)(gtp + NLD * 1024));
      __builtin_nontemporal_store(vp, (v2dd*)(gpp + NLD * 1024));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the region has been read out before the next tile lands in it
    row += step;
    n += step_n;
    rem += step_r;
    if (rem >= d) {
      rem -= d;
      n += 1;
    }
  }
  // the ragged last tile: the wavefront whose turn it would be works straight on memory
  if (nfull * 64 < rows_total && nfull % W == w0) {
    const int64_t r = nfull * 64 + lane;
    if (r < rows_total) {
      const double* rt = a.theta + r * S;
      const double* rp = a.phi + r * S;
      double* ot = a.gtheta + r * S;
      double* op = a.gphi + r * S;
      row_grads(
          a.gx[r] * ilt_row_scale(sc_args, a.t[r / a.d]),
          [&](int k, double* th, double* ph) {
            *th = rt[k];
            *ph = rp[k];
          },
          [&](int k, double gt, double gp) {
            ot[k] = gt;
            op[k] = gp;
          });
    }
  }
}
template <int S>
static bool launch_bwd_rows_instance(const IltBwdArgs& a, hipStream_t s, hipError_t* err) {
  constexpr int SLOT = (64 * S * 8 + 1023) / 1024 * 1024;
  constexpr size_t shmem = (size_t)4 * 2 * SLOT;
  static std::atomic<unsigned long long> attr_done{0};
  if (!lds_attr_once(ilt_fourier_bwd_rows_kernel<S>, shmem, attr_done)) return false;
  const int64_t tiles = (a.N * a.d + 63) / 64;
  const int per_cu = S <= 17 ? 2 : 1;
  int64_t grid = (tiles + 3) / 4;
  if (grid > 256 * per_cu) grid = 256 * per_cu;
  hipLaunchKernelGGL((ilt_fourier_bwd_rows_kernel<S>), dim3((unsigned)grid), dim3(256), shmem, s, a);
  *err = hipGetLastError();
  return true;
}
static bool launch_ilt_fourier_bwd_rows(const IltBwdArgs& a, hipStream_t s, hipError_t* err) {
  if (a.S < 3 || a.S > 33 || (a.S & 1) == 0 || a.scale != 2.0) return false;
  if ((((uintptr_t)a.theta) | ((uintptr_t)a.phi) | ((uintptr_t)a.gtheta) | ((uintptr_t)a.gphi)) & 15) return false;
#if NLC_ILT_EXPERIMENTS
  static const int rows_env = [] {
    const char* ev = std::getenv("NLC_ILT_ROWS");
    return ev ? std::atoi(ev) : 1;
  }();
  if (!rows_env) return false;
#endif
  switch (a.S) {
#define NLC_ROWS_CASE(SS) \
  case SS:                \
    return launch_bwd_rows_instance<SS>(a, s, err);
    NLC_ROWS_CASE(3) NLC_ROWS_CASE(5) NLC_ROWS_CASE(7) NLC_ROWS_CASE(9) NLC_ROWS_CASE(11) NLC_ROWS_CASE(13) NLC_ROWS_CASE(15)
    NLC_ROWS_CASE(17) NLC_ROWS_CASE(19) NLC_ROWS_CASE(21) NLC_ROWS_CASE(23) NLC_ROWS_CASE(25) NLC_ROWS_CASE(27) NLC_ROWS_CASE(29)
    NLC_ROWS_CASE(31) NLC_ROWS_CASE(33)
#undef NLC_ROWS_CASE
  }
  return false;
}

hipError_t launch_ilt_fourier_bwd(const IltBwdArgs& a_in, hipStream_t s) {
  IltBwdArgs a = a_in;
  const int64_t rows_total = a.N * a.d;
  if (rows_total <= 0) return hipSuccess;
  if (a.S > 256) return hipErrorInvalidValue;
  {
    hipError_t e2 = hipSuccess;
    if (launch_ilt_fourier_bwd_rows(a, s, &e2)) return e2;
  }
  a.rpp = 256 / a.S;
  if (a.rpp > 32) a.rpp = 32;
  a.iters = 256 / a.rpp / 8 * 8;  // rows = rpp * iters <= 256: one thread per row stages the row's gradient
  if (a.iters < 8) return hipErrorInvalidValue;
  const int rows = a.rpp * a.iters;
  const int64_t nblk = (rows_total + rows - 1) / rows;
  const unsigned grid = (unsigned)(nblk < 4096 ? nblk : 4096);
  hipLaunchKernelGGL(ilt_fourier_bwd_kernel, dim3(grid), dim3(256), (size_t)rows * sizeof(double), s, a);
  return hipGetLastError();
}

}  // namespace nlc
