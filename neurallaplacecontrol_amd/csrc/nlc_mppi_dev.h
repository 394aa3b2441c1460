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

// One weight block = kWeightBlockSamples consecutive samples of episode e, handled by a 256-thread workgroup:
// w_k = exp(-(c_k - beta)/lambda) -> cost_nz; partial eta and S[t,j] = sum_k w_k eps[k,t,j] -> block_part.  Thread tj
// < T*nu walks the block's samples with coalesced reads of the eps[k, :, :] rows, the weights broadcast from LDS.
// `sw`: kWeightBlockSamples doubles of LDS.  M: how cost / noise are loaded and block_part is stored (see above).
template <class M>
__device__ __forceinline__ void weight_block(const WeightArgs& a, int e, int blk, double beta, double* sw) {
#pragma clang fp contract(off)
  const int64_t kb = (int64_t)blk * kWeightBlockSamples;  // within the episode
  const int ns = (int)((a.Kep - kb < kWeightBlockSamples) ? (a.Kep - kb) : kWeightBlockSamples);
  const int64_t k0 = (int64_t)e * a.Kep + kb;
  const int TN = a.T * a.nu;
  double wk = 0.0;
  if ((int)threadIdx.x < ns) {
    wk = exp(-(1.0 / a.lambda_) * (M::ld(a.cost + k0 + threadIdx.x) - beta));  // _ensure_non_zero :12-13
    a.cost_nz[k0 + threadIdx.x] = wk;
    sw[threadIdx.x] = wk;
  }
  __syncthreads();
  double* out = a.block_part + ((int64_t)e * a.nblk + blk) * (1 + TN);
  if (threadIdx.x < 64) {
    const double es = wave_sum(wk);  // threads 0..63 hold all (<= 64) weights of the block
    if (threadIdx.x == 0) {
      M::st(out, es);
      if (blk == 0) a.partials[(int64_t)e * (2 + TN)] = beta;
    }
  }
  for (int tj = threadIdx.x; tj < TN; tj += 256) {
    double acc = 0.0;
    const double* np = a.noise + k0 * TN + tj;
    for (int s = 0; s < ns; ++s) acc += sw[s] * M::ld(np + (int64_t)s * TN);
    M::st(out + 1 + tj, acc);
  }
}

// Entry i (0: eta, 1 + tj: S[tj]) of the shard's partials: the block partials folded in a fixed order (run-to-run
// deterministic) by ONE wavefront -- lanes stride over the blocks, then a wave reduction.
template <class M>
__device__ __forceinline__ void weight_final_entry(const WeightArgs& a, int64_t e, int i, int lane) {
#pragma clang fp contract(off)
  const int TN = a.T * a.nu;
  double acc = 0.0;
  for (int b = lane; b < a.nblk; b += 64) acc += M::ld(a.block_part + (e * a.nblk + b) * (1 + TN) + i);
  acc = wave_sum(acc);
  if (lane == 0) a.partials[e * (2 + TN) + 1 + i] = acc;
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
