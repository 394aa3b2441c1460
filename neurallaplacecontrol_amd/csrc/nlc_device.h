// Device-side helpers shared by the NLC kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "nlc_math.h"

namespace nlc {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr double kPi = 3.14159265358979323846;

// v_mfma_f64_16x16x4_f64: D(16x16) = A(16x4) * B(4x16) + C, one f64 of A and of B per lane.
//   A: lane l holds A[m = l & 15][k = l >> 4]
//   B: lane l holds B[k = l >> 4][n = l & 15]
//   D: lane l, reg r holds D[row = (l >> 4) + 4 r][col = l & 15]
// Used with WEIGHTS as A (rows = output features) and SAMPLES as B/D columns, so accumulator register
// r of output tile j (feature 16 j + 4 r + q) is directly the B fragment of k-step 4 j + r of the next
// layer: layers chain with no lane movement and no LDS round trip.
__device__ __forceinline__ v4d mfma(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ v4d splat(double x) { return v4d{x, x, x, x}; }


// Make a (wave-uniform) pointer opaque to the optimiser.  Without this LICM hoists the hundreds of
// "base + constant" fragment addresses of an unrolled GEMM out of the enclosing loop and spills them.
typedef const __attribute__((address_space(1))) double* gptr;  // global (HBM) address space
__device__ __forceinline__ gptr opaque(gptr p) {
  asm volatile("" : "+s"(p));
  return p;
}
__device__ __forceinline__ gptr opaque(const double* p) { return opaque((gptr)p); }
// the same for a table a workgroup keeps in LDS (per-lane address in a VGPR)
typedef const __attribute__((address_space(3))) double* lptr;
__device__ __forceinline__ lptr opaque_lds(lptr p) {
  asm volatile("" : "+v"(p));
  return p;
}
__device__ __forceinline__ lptr opaque(lptr p) { return opaque_lds(p); }

// NLC_GEMM_INTERLEAVE: issue order of one k-step -- 0: the MT fragment loads of the next k-step, then the MT MFMAs (rounds 1-3);
// 1: MFMA, load, MFMA, load, ... (sched_group_barrier), every load issued in the shadow of an MFMA.  A vector load costs the
// issuing wave ~17 clocks; a burst of MT of them in front of the k-step's first MFMA leaves the matrix pipe idle that long
// when the SIMD holds one wave (round 4, nl_rollout_kernel at K = 16384: 1.240 -> 1.175 ms; profiles/r4_rollout_phase_clocks.md).
#ifndef NLC_GEMM_INTERLEAVE
#define NLC_GEMM_INTERLEAVE 1
#endif
// (E: further loads the k-step carries for a LATER phase -- gemm_acc_head's `extra` -- spread over the same shadows; FRAGS = 0:
// the last k-step, which fetches no fragments of its own)
template <int MT, int E = 0, int FRAGS = 1, bool LDSW = false>
__device__ __forceinline__ void gemm_kstep_order() {
#if NLC_GEMM_INTERLEAVE
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    constexpr int kBase = E / MT, kRem = E % MT;
    const int nv = FRAGS + kBase + (m < kRem ? 1 : 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
    constexpr int kRd = LDSW ? 0x100 : 0x020;  // the fragments' kind: DS read / VMEM read
    if (nv == 1) __builtin_amdgcn_sched_group_barrier(kRd, 1, 0);  // reads in its shadow
    if (nv == 2) __builtin_amdgcn_sched_group_barrier(kRd, 2, 0);
    if (nv == 3) __builtin_amdgcn_sched_group_barrier(kRd, 3, 0);
    if (nv == 4) __builtin_amdgcn_sched_group_barrier(kRd, 4, 0);
    if (nv >= 5) __builtin_amdgcn_sched_group_barrier(kRd, 5, 0);
  }
#endif
}

// acc[m] (+)= sum_ks  Wp[ks][m] (x) bfrag(ks).  Wp is fragment-packed: Wp[(ks*MT + m)*64 + lane],
// wave-uniform base (SGPR) + lane offset.  A fragments are prefetched one k-step ahead so the L2/L1
// latency hides under the MT MFMAs of the current step.
template <int MT, int KS, typename BF>
__device__ __forceinline__ void gemm_acc(v4d (&acc)[MT], const double* __restrict__ Wp, int lane, BF bfrag) {
  double a_cur[MT], a_nxt[MT];
  gptr p = opaque(Wp);
#pragma unroll
  for (int m = 0; m < MT; ++m) a_cur[m] = p[m * 64 + lane];
  // B fragments are produced one k-step ahead as well: when bfrag() is an activation (tanh of the previous
  // layer's accumulator) its VALU work sits next to independent MFMAs and hides in their shadow
  double b_cur = bfrag(0), b_nxt = 0.0;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) {
      p = opaque(p + MT * 64);
#pragma unroll
      for (int m = 0; m < MT; ++m) a_nxt[m] = p[m * 64 + lane];
      b_nxt = bfrag(ks + 1);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = mfma(a_cur[m], b_cur, acc[m]);
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
    b_cur = b_nxt;
    gemm_kstep_order<MT>();
    // keep the scheduler from hoisting every k-step's fragment loads to the top
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same in two calls and for MT consecutive output tiles m0 .. m0 + MT - 1 of a matrix packed with MTOT tiles per k-step (a
// layer's output tiles may be processed in two passes: half the accumulators live at a time), so that a GEMM's first fragments
// can be in flight during the phase BEFORE it (activation, epilogue): gemm_head issues the loads of k-step 0, gemm_acc_head runs
// the k loop from them.  Per tile the MFMA sequence over k is gemm_acc's: same bits.
// (WP: const double* / gptr -- the packed matrix in HBM / L2 -- or lptr, a copy in LDS)
template <int MT, class WP>
__device__ __forceinline__ void gemm_head(double (&a0)[MT], WP Wp, int m0, int lane) {
  const auto p = opaque(Wp + (size_t)m0 * 64);
#pragma unroll
  for (int m = 0; m < MT; ++m) a0[m] = p[m * 64 + lane];
}
// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}) -- every index a constant
// expression, so register arrays written through it never need a run-time subscript (which would put them in scratch)
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// extra(ks) (EPK loads per k-step, any phase's: bias tiles, the next GEMM's first fragments, coefficient tables) is issued with
// the k-step's own fragment loads and scheduled into the MFMA shadows with them: on this chip a vector load costs its wave
// ~17 issue clocks wherever it is NOT behind an MFMA.
struct NoExtra {
  template <class KC>
  __device__ __forceinline__ void operator()(KC) const {}
};
// NLC_GEMM_PD: prefetch distance of the k loop in k-steps (the fragments of k-step ks + PD are issued during k-step ks).
// Measured at the headline shape (round 4): 1 -> 1.183 ms, 2 -> 1.200, 3 -> 1.221: the fragments come back in time, more of
// them in flight only costs.
#ifndef NLC_GEMM_PD
#define NLC_GEMM_PD 1
#endif
template <int MT, int MTOT, int KS, int EPK = 0, class WP, typename BF, typename EX = NoExtra>
__device__ __forceinline__ void gemm_acc_head(v4d (&acc)[MT], const double (&a0)[MT], WP Wp, int m0, int lane, BF bfrag,
                                              EX extra = EX{}) {
  constexpr bool LDSW = std::is_same<WP, lptr>::value;
  constexpr int PD = NLC_GEMM_PD < KS ? NLC_GEMM_PD : 1;
  double a[PD + 1][MT];  // ring of fragment sets: k-step ks lives in a[ks % (PD + 1)]
  // HBM / L2: a wave-uniform base advanced per k-step (SALU) + immediate tile offsets.  LDS: ONE per-lane base for the whole
  // matrix, every fragment at a compile-time offset from it (ds_read2st64_b64 reaches 255 tiles of 512 B), no address arithmetic.
  auto p = opaque(Wp + (size_t)m0 * 64);
  auto frag = [&](auto kc, int m) {
    constexpr int k = decltype(kc)::value;
    if constexpr (LDSW)
      return p[(k * MTOT + m) * 64 + lane];
    else
      return p[m * 64 + lane];
  };
  auto advance = [&]() {
    if constexpr (!LDSW) p = opaque(p + MTOT * 64);
  };
#pragma unroll
  for (int m = 0; m < MT; ++m) a[0][m] = a0[m];
  // k-steps 1 .. PD - 1 before the loop (PD = 1: none)
  static_for<PD - 1>([&](auto jc) {
    constexpr int j = decltype(jc)::value + 1;
    advance();
#pragma unroll
    for (int m = 0; m < MT; ++m) a[j][m] = frag(std::integral_constant<int, j>{}, m);
  });
  double b_cur = bfrag(0), b_nxt = 0.0;
  static_for<KS>([&](auto ks_c) {
    constexpr int ks = decltype(ks_c)::value;
    if constexpr (ks + PD < KS) {
      advance();
#pragma unroll
      for (int m = 0; m < MT; ++m) a[(ks + PD) % (PD + 1)][m] = frag(std::integral_constant<int, ks + PD>{}, m);
    }
    if (ks + 1 < KS) b_nxt = bfrag(ks + 1);
    extra(ks_c);  // (ks as a compile-time constant)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = mfma(a[ks % (PD + 1)][m], b_cur, acc[m]);
    b_cur = b_nxt;
    if (ks + PD < KS)
      gemm_kstep_order<MT, EPK, 1, LDSW>();
    else
      gemm_kstep_order<MT, EPK, 0, LDSW>();
    __builtin_amdgcn_sched_barrier(0);
  });
}

// bias tile: lane (q = lane >> 4) reg r holds feature 16 j + 4 r + q
__device__ __forceinline__ v4d load_bias_tile(const double* __restrict__ b, int j, int q) {
  const double* p = b + 16 * j + q;
  return v4d{p[0], p[4], p[8], p[12]};
}

// ---- Philox4x32-10 (Salmon et al. 2011), counter-based so a K-shard reproduces the unsharded stream
struct u4 {
  uint32_t x, y, z, w;
};
__host__ __device__ inline u4 philox4x32_10(u4 ctr, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)M0 * ctr.x, p1 = (uint64_t)M1 * ctr.z;
    u4 n;
    n.x = (uint32_t)(p1 >> 32) ^ ctr.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ ctr.w ^ k1;
    n.w = (uint32_t)p0;
    ctr = n;
    k0 += W0;
    k1 += W1;
  }
  return ctr;
}
// two uint32 -> uniform double in (0,1): 53 random bits, never 0
__host__ __device__ inline double u53(uint32_t hi, uint32_t lo) {
  const uint64_t v = (((uint64_t)hi << 32) | lo) >> 11;
  return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}

}  // namespace nlc
