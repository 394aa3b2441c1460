// Device-side helpers shared by the NLC kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
    // keep the scheduler from hoisting every k-step's fragment loads to the top
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same for MT consecutive output tiles m0 .. m0 + MT - 1 of a matrix packed with MTOT tiles per k-step (a layer's output
// tiles processed in two halves: half the accumulators and half the prefetch registers live at a time; per tile the
// MFMA sequence over k is unchanged, so the results are the same bits).
template <int MT, int MTOT, int KS, typename BF>
__device__ __forceinline__ void gemm_acc_part(v4d (&acc)[MT], const double* __restrict__ Wp, int m0, int lane, BF bfrag) {
  double a_cur[MT], a_nxt[MT];
  gptr p = opaque(Wp + (size_t)m0 * 64);
#pragma unroll
  for (int m = 0; m < MT; ++m) a_cur[m] = p[m * 64 + lane];
  double b_cur = bfrag(0), b_nxt = 0.0;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) {
      p = opaque(p + MTOT * 64);
#pragma unroll
      for (int m = 0; m < MT; ++m) a_nxt[m] = p[m * 64 + lane];
      b_nxt = bfrag(ks + 1);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = mfma(a_cur[m], b_cur, acc[m]);
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
    b_cur = b_nxt;
    __builtin_amdgcn_sched_barrier(0);
  }
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
