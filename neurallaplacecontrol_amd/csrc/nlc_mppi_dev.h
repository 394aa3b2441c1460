// Device bodies of the MPPI sampling / bounding / importance-weighting steps (planners/mppi_delay.py:210-216, 319-328),
// shared by the stand-alone kernels (kernels_mppi.hip) and the one-launch planner body (kernels_fused.hip), which runs
// them as phases of its persistent grid.  Both translation units must produce the SAME bits (the fused body is tested
// bit for bit against the launch-per-step path), and they are built with different -ffp-contract defaults, so every
// function here pins `fp contract(off)` in its own block: a*b+c is a multiply and an add, as in the torch-CPU op order.
// (nlc_math.h's kernels spell their FMAs out, so they do not depend on the setting either.)
#pragma once
#include "nlc_device.h"
#include "nlc_kernels.h"

namespace nlc {

// ------------------------------------------------------------------ memory access flavours
// Plain: data written by an EARLIER launch.  Sc1: data handed over INSIDE a launch -- write-through stores, L1-bypassing
// loads (cdna_hip_programming.md Guideline 16; 8-byte agent-scope relaxed atomics lower to global_load/store_dwordx2 sc1).
struct MemPlain {
  __device__ __forceinline__ static double ld(const double* p) { return *p; }
  __device__ __forceinline__ static void st(double* p, double v) { *p = v; }
};
struct MemSc1 {
  __device__ __forceinline__ static double ld(const double* p) {
    return __builtin_bit_cast(double, __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  __device__ __forceinline__ static void st(double* p, double v) {
    __hip_atomic_store((unsigned long long*)p, __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
};

// ------------------------------------------------------------------ sampling (:319-328)
// eps ~ N(mu, Sigma) for global sample kg, horizon step t of command `counter`: Philox4x32-10 keyed by the seed, one block
// = two normals (Box-Muller), coloured by the lower Cholesky factor.  The draw does not depend on the sharding.
__device__ __forceinline__ void mppi_draw(int64_t kg, int t, uint64_t seed, uint64_t counter, int nu, const double* mu,
                                          const double* chol, double (&eps)[NLC_MAX_NU]) {
#pragma clang fp contract(off)
  static_assert(NLC_MAX_NU <= 2, "one Philox block yields two normals");
  double z[NLC_MAX_NU];
  const u4 r = philox4x32_10(u4{(uint32_t)kg, (uint32_t)(kg >> 32), (uint32_t)t, (uint32_t)counter}, (uint32_t)seed,
                             (uint32_t)(seed >> 32) ^ (uint32_t)(counter >> 32));
  const double u1 = u53(r.x, r.y), u2 = u53(r.z, r.w);
  const double rad = sqrt(-2.0 * log(u1));
  double sn, cs;
  m::sincos_bounded(2.0 * kPi * u2 - kPi, &sn, &cs);  // angle in (-pi, pi)
  z[0] = rad * cs;
  z[1] = rad * sn;
  // (fixed trip counts: the small arrays stay in registers)
#pragma unroll
  for (int i = 0; i < NLC_MAX_NU; ++i) {
    eps[i] = 0.0;
    if (i < nu) {
      double v = mu[i];
#pragma unroll
      for (int j = 0; j < NLC_MAX_NU; ++j)
        if (j <= i) v += chol[i * nu + j] * z[j];
      eps[i] = v;
    }
  }
}

// U <- roll(U, -1); U[-1] = u_init (:199-200), read on the fly from the sequence BEFORE the shift
__device__ __forceinline__ double mppi_shifted_U(const double* U_old, const double* u_init, int64_t e, int T, int nu, int t,
                                                 int i) {
  return (t + 1 < T) ? U_old[(e * T + t + 1) * nu + i] : u_init[i];
}

// perturbed action V = bound(U + eps) in normalised units (:322-326); the bounded noise is V - U (:328)
__device__ __forceinline__ double mppi_bound(double U, double eps, bool null_action, double u_scale, int has_bounds,
                                             double u_min, double u_max) {
#pragma clang fp contract(off)
  double V = U + eps;
  if (null_action) V = 0.0;  // :322-323
  double Vs = V * u_scale;
  if (has_bounds) Vs = fmax(fmin(Vs, u_max), u_min);  // :351
  return Vs / u_scale;                                // :326
}

// ------------------------------------------------------------------ importance weights (:210-216)
__device__ __forceinline__ double wave_min(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma clang fp contract(off)
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Two levels, so that a 16-sample rollout tile can fold its own samples the moment its costs are final (the fused planner
// body does, inside its launch) and every path produces the same bits:
//   tile b (kWeightTile consecutive samples of episode e):   beta_b = min c,  w_s = exp(-(c_s - beta_b)/lambda),
//                                                            eta_b = sum w_s,  S_b[t,j] = sum_s w_s eps[s,t,j]
//   rank (all nblk tiles of the episode):  beta = min beta_b,  scale_b = exp(-(beta_b - beta)/lambda),
//                                          eta = sum scale_b eta_b,  S = sum scale_b S_b          -> partials (beta, eta, S)
// -- the same rescaling the shard merge applies between ranks.  cost_nz / omega come from merge_kernel.
// tile_part layout: (E, nblk, 2 + T*nu) = (beta_b, eta_b, S_b[T*nu]).
//
// weight_tile: ONE wavefront.  `cost`: the cost of sample (lane & 15) of the tile in every lane (any value where !valid).
template <class M>
__device__ __forceinline__ void weight_tile(const WeightArgs& a, int64_t e, int64_t b, int lane, double cost, bool valid) {
#pragma clang fp contract(off)
  const int TN = a.T * a.nu;
  const int64_t k0 = e * a.Kep + b * kWeightTile;              // first sample of the tile (flat index)
  const int64_t left = a.Kep - b * kWeightTile;
  const int ns = (int)(left < kWeightTile ? left : kWeightTile);
  double beta = valid ? cost : INFINITY;
  for (int o = 8; o > 0; o >>= 1) beta = fmin(beta, __shfl_xor(beta, o, 64));
  const double w = valid ? exp(-(1.0 / a.lambda_) * (cost - beta)) : 0.0;  // _ensure_non_zero :12-13
  double eta = w;
  for (int o = 8; o > 0; o >>= 1) eta += __shfl_xor(eta, o, 64);
  double* out = a.tile_part + (e * a.nblk + b) * (2 + TN);
  if (lane == 0) {
    M::st(out, beta);
    M::st(out + 1, eta);
  }
  for (int tj0 = 0; tj0 < TN; tj0 += 64) {
    const int tj = tj0 + lane;
    const bool on = tj < TN;
    double acc = 0.0;
#pragma unroll 1
    for (int s0 = 0; s0 < kWeightTile; s0 += 8) {
      double ev[8];  // eight loads in flight (M may be an ordered access), then their multiply-adds in sample order
#pragma unroll
      for (int s = 0; s < 8; ++s) ev[s] = (on && s0 + s < ns) ? M::ld(a.noise + (k0 + s0 + s) * TN + tj) : 0.0;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const double ws = __shfl(w, s0 + s, 64);
        if (s0 + s < ns) acc += ws * ev[s];
      }
    }
    if (on) M::st(out + 2 + tj, acc);
  }
}

// Rank fold.  Fixed order (independent of the launch shape): tiles in CHUNKS of 64; within a chunk wavefront w adds the 16
// tiles 16 w .. 16 w + 15 in ascending order (lane i owning entry i: 0 eta, 1 + tj S[tj]; loads issued eight at a time), the chunk's value is ((s0 + s1) + s2) + s3, and the chunks are added in ascending
// order.  weight_chunk: one 256-thread workgroup, one chunk, entries i0 .. i0 + 63; returns the chunk's value in the lanes
// of wavefront 0.  `lds`: kWeightRankLds doubles.  beta: the minimum over ALL tiles of the episode.
constexpr int kWeightChunk = 64;
constexpr int kWeightRankLds = 64 + 8 + 4 * 64;
__host__ __device__ inline int weight_chunks(int nblk) { return (nblk + kWeightChunk - 1) / kWeightChunk; }
template <class M>
__device__ __forceinline__ double weight_chunk(const WeightArgs& a, int64_t e, int j, int i0, double beta, double* lds) {
#pragma clang fp contract(off)
  const int TN = a.T * a.nu, W = 2 + TN;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double* part = a.tile_part + e * a.nblk * W;
  double* s_scale = lds;       // [64] exp(-(beta_b - beta)/lambda) of the chunk's tiles
  double* s_acc = lds + 72;    // [4][64]
  const int i = i0 + lane;
  const bool on = i < 1 + TN;
  __syncthreads();  // (the previous chunk's s_scale / s_acc have been read)
  if (tid < kWeightChunk && j * kWeightChunk + tid < a.nblk)
    s_scale[tid] = exp(-(1.0 / a.lambda_) * (M::ld(part + (int64_t)(j * kWeightChunk + tid) * W) - beta));
  __syncthreads();
  const int b0 = j * kWeightChunk + 16 * wv;
  double acc = 0.0;
#pragma unroll 1
  for (int s0 = 0; s0 < 16; s0 += 8) {
    double v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = (on && b0 + s0 + s < a.nblk) ? M::ld(part + (int64_t)(b0 + s0 + s) * W + 1 + i) : 0.0;
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (b0 + s0 + s < a.nblk) acc += s_scale[16 * wv + s0 + s] * v[s];
  }
  s_acc[wv * 64 + lane] = acc;
  __syncthreads();
  return ((s_acc[lane] + s_acc[64 + lane]) + s_acc[128 + lane]) + s_acc[192 + lane];
}
// minimum of the tile minima of episode e (exact, so its order is free); `lds`: 4 doubles at lds + 64
template <class M>
__device__ __forceinline__ double weight_beta(const WeightArgs& a, int64_t e, double* lds) {
  const int W = 2 + a.T * a.nu;
  const int tid = threadIdx.x;
  const double* part = a.tile_part + e * a.nblk * W;
  double bmin = INFINITY;
  for (int b = tid; b < a.nblk; b += 256) bmin = fmin(bmin, M::ld(part + (int64_t)b * W));
  bmin = wave_min(bmin);
  __syncthreads();
  if ((tid & 63) == 0) lds[64 + (tid >> 6)] = bmin;
  __syncthreads();
  return fmin(fmin(lds[64], lds[65]), fmin(lds[66], lds[67]));
}
// weight_rank: ONE workgroup folds every chunk of episode e into the shard's partials (the fused planner body's last
// rollout tile; small populations).  Same arithmetic as weight_chunk_kernel + weight_final_kernel.
template <class M>
__device__ __forceinline__ void weight_rank(const WeightArgs& a, int64_t e, double* lds) {
#pragma clang fp contract(off)
  const int TN = a.T * a.nu, W = 2 + TN;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double beta = weight_beta<M>(a, e, lds);
  const int nch = weight_chunks(a.nblk);
  for (int i0 = 0; i0 < 1 + TN; i0 += 64) {
    double tot = 0.0;
    for (int j = 0; j < nch; ++j) tot += weight_chunk<M>(a, e, j, i0, beta, lds);
    if (wv == 0 && i0 + lane < 1 + TN) a.partials[e * W + 1 + i0 + lane] = tot;
  }
  if (threadIdx.x == 0) a.partials[e * W] = beta;
}

// order-preserving map double -> uint64 (a < b  <=>  key(a) < key(b)) and back: min over doubles by integer atomics.
// The fused body keeps max(~key(cost)), which starts from the zeroed word.
__device__ __forceinline__ unsigned long long f64_order_key(double x) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_from_order_key(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __builtin_bit_cast(double, b);
}

}  // namespace nlc
