// One row of the de Hoog, Knight & Stokes accelerated inverse Laplace transform: the quotient-difference table of the 2M + 1
// terms a_0 .. a_2M of one (point, dim) row, its continued-fraction coefficients and the A / B recurrence with the improved
// remainder -- restated from mpmath 1.3.0 calculus/inverselaplace.py:476-531 (see kernels_dehoog.hip for the formulation and
// its measurements).  Shared by the stand-alone kernel (kernels_dehoog.hip, three input layouts) and the persistent step
// chain of the de Hoog planner (kernels_dehoog_chain.hip): ONE implementation, so every path produces the same bits.
#pragma once
#include "nlc_cplx.h"
#include "nlc_device.h"

namespace nlc {

// Term source over SLOT-major F: term k of this lane's row is fre / fim[slot_of_term[k] * stride + col] -- the lanes of a
// wavefront read consecutive columns, so every load is one full line.  CH terms are fetched at a time into registers.
template <int CH>
struct DehoogSlotTerms {
  const double* fre;
  const double* fim;
  const int* slot_of_term;  // wave-uniform: (S) slots of this row's dim
  int64_t stride, col;
  cplx buf[CH];
  template <int S>
  __device__ __forceinline__ void stage(int n) {
    const int nt = (S - n < CH) ? (S - n) : CH;
#pragma clang loop unroll(full)
    for (int k = 0; k < CH; ++k)
      if (k < nt) {
        const int64_t at = (int64_t)slot_of_term[n + k] * stride + col;
        buf[k] = {fre[at], fim[at]};
      }
  }
  __device__ __forceinline__ cplx term(int n) const { return buf[n % CH]; }
};

// The same over a per-LANE slot table (the lanes of one wavefront may belong to two dims: kernels_dehoog_chain.hip with 32-sample
// blocks): slot_of_term is this lane's dim's (S) table, read with vector loads.
template <int CH>
struct DehoogSlotTermsLane {
  const double* fre;
  const double* fim;
  const int* slot_of_term;  // per lane
  int64_t stride, col;
  cplx buf[CH];
  template <int S>
  __device__ __forceinline__ void stage(int n) {
    const int nt = (S - n < CH) ? (S - n) : CH;
#pragma clang loop unroll(full)
    for (int k = 0; k < CH; ++k)
      if (k < nt) {
        const int64_t at = (int64_t)slot_of_term[n + k] * stride + col;
        buf[k] = {fre[at], fim[at]};
      }
  }
  __device__ __forceinline__ cplx term(int n) const { return buf[n % CH]; }
};

// diagonals per pass of dehoog_row (below): where a wavefront has its SIMD to itself / where two share one
// (tools A/B: -DNLC_DEHOOG_SKEW_ALONE=n / -DNLC_DEHOOG_SKEW_SHARED=n, n = 1, 2, 4)
#ifndef NLC_DEHOOG_SKEW_ALONE
#define NLC_DEHOOG_SKEW_ALONE 4
#endif
#ifndef NLC_DEHOOG_SKEW_SHARED
#define NLC_DEHOOG_SKEW_SHARED 2
#endif
constexpr int kDehoogSkewAlone = NLC_DEHOOG_SKEW_ALONE, kDehoogSkewShared = NLC_DEHOOG_SKEW_SHARED;

// SRC: stage<S>(n) is called before term n whenever n % CH == 0 (it makes terms [n, n + CH) available), term(n) returns a_n.
// Returns A_2M / B_2M, the continued fraction with the improved remainder; the caller scales Re by e^{gamma t} / T.
//
// Round 5: W anti-diagonals per pass.  Along a diagonal every entry depends on the one before it (column c on column c - 1), so a
// lone wavefront walked a chain of ~530 dependent complex divisions / sums per row at the FP64 pipe's dependent-issue latency -- the
// planner's QD launch (640 wavefronts on 1 024 SIMDs) is exactly one such chain long, 25 us.  Entry (n + 1, c) needs (n + 1, c - 1),
// (n, c - 1) and (n, c - 2) only, so diagonal n + 1 can run ONE column behind diagonal n, n + 2 one behind that, ...: a pass issues
// column c of diagonal n, c - 1 of n + 1, ..., c - W + 1 of n + W - 1 together -- W independent chains, the same operations on the
// same operands in every entry (bit-identical for every W), D[] still updated in place (diagonal n + j reads what n + j - 1 wrote one
// step earlier).  W = 2 where two wavefronts share a SIMD anyway (the stand-alone row kernels: +12 VGPRs per chain must stay
// under 256), W = 4 where a wavefront is alone (the planner's slot-major launch, the persistent chain kernel).  W = 1: rounds 1-4.
template <int M, int CH, int W = kDehoogSkewShared, class SRC>
__device__ __forceinline__ cplx dehoog_row(SRC& src, const cplx z) {
#pragma clang fp contract(off)
  constexpr int S = 2 * M + 1;
  cplx D[2 * M];
  cplx a_prev = {0.0, 0.0}, d0 = {0.0, 0.0};
  // A/B continued-fraction recurrence, fed with d_1, d_2, ... as the diagonals produce them
  cplx A_prev = {0.0, 0.0}, A_cur = {0.0, 0.0}, B_prev = {1.0, 0.0}, B_cur = {1.0, 0.0};
  cplx d_last = {0.0, 0.0}, d_cur = {0.0, 0.0};
  auto feed = [&](int n, const cplx last) {  // d_n = -(entry at i = 0); d_2M only enters the remainder
    d_last = d_cur;
    d_cur = {-last.re, -last.im};
    if (n != 2 * M) {
      const cplx dz = cmul(d_cur, z);
      const cplx An = cadd(A_cur, cmul(dz, A_prev));
      const cplx Bn = cadd(B_cur, cmul(dz, B_prev));
      A_prev = A_cur;
      A_cur = An;
      B_prev = B_cur;
      B_cur = Bn;
    }
  };
  // one entry of a diagonal: column c (c = 3: q_2, 4: e_2, ...) from the diagonal's previous entry `newv` and the previous
  // diagonal's columns c - 1 (`old1`) and c - 2 (`old2`); D[c - 1] is replaced in place
  auto step = [&](int c, cplx& newv, cplx& old1, cplx& old2) {
    const cplx oldc = D[c - 1];
    const cplx val = (c & 1) ? cdiv(cmul(old2, newv), old1) : cadd(csub(newv, old1), old2);
    D[c - 1] = val;
    old2 = old1;
    old1 = oldc;
    newv = val;
  };
  // diagonals n .. n + WW - 1 in one pass (WW a compile-time width, every index below a constant of the unrolled code)
  auto pass = [&](int n, auto ww) {
    constexpr int WW = decltype(ww)::value;
    cplx v[WW], o1[WW], o2[WW];
    // column 1 of each: q_1^(n+j-1) = a_(n+j) / a_(n+j-1); a diagonal's "previous diagonal" is the one before it in the pass
#pragma unroll
    for (int j = 0; j < WW; ++j) {
      if ((n + j) % CH == 0) src.template stage<S>(n + j);  // (earlier terms of the pass are in registers already)
      const cplx an = src.term(n + j);
      v[j] = cdiv(an, a_prev);
      a_prev = an;
      o1[j] = j == 0 ? D[0] : v[j - 1];
      o2[j] = {0.0, 0.0};  // column 0: e_0 = 0
    }
    D[0] = v[WW - 1];
    // time step s: diagonal n + j works on column s - j (columns 2 .. n + j); diagonal n + j reads D[s - j - 1] as diagonal
    // n + j - 1 left it at step s - 1, and its own store does not touch what the diagonals ahead of it read in this step
#pragma unroll
    for (int s = 2; s <= n + 2 * (WW - 1); ++s) {
#pragma unroll
      for (int j = 0; j < WW; ++j) {
        const int c = s - j;
        if (c >= 2 && c <= n + j) step(c, v[j], o1[j], o2[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < WW; ++j) feed(n + j, v[j]);
    // keep the passes apart: hoisting the next terms' reads / interleaving further diagonals only costs registers
    // (the asm ties this pass's results -- including BOTH continued-fraction recurrences, which the compiler
    // otherwise defers to the end of the kernel, spilling every d_n z to scratch: 1 GB of HBM writes per launch --
    // to a memory barrier, so the arithmetic cannot sink below the following reads either)
    asm volatile(""
                 : "+v"(A_cur.re), "+v"(A_cur.im), "+v"(B_cur.re), "+v"(B_cur.im), "+v"(d_cur.re), "+v"(d_cur.im)::"memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    src.template stage<S>(0);
    const cplx a0 = src.term(0);
    d0 = {0.5 * a0.re, 0.5 * a0.im};  // a_0 enters halved
    a_prev = d0;
    A_cur = d0;
  }
  constexpr int kFull = (2 * M) / W, kRest = (2 * M) % W;  // 2M is even: the rest is 0 for W = 1, 2 and 0 or 2 for W = 4
  static_assert(W == 1 || W == 2 || W == 4, "dehoog_row: 1, 2 or 4 diagonals per pass");
#pragma clang loop unroll(full)
  for (int p = 0; p < kFull; ++p) pass(1 + p * W, std::integral_constant<int, W>{});
  if constexpr (kRest == 2) pass(1 + kFull * W, std::integral_constant<int, 2>{});
  // here d_last = d_{2M-1}, d_cur = d_{2M}; the recurrence has run for i = 1 .. 2M-1
  const cplx diff = csub(d_last, d_cur);
  const cplx one = {1.0, 0.0};
  cplx brem = cadd(one, cmul(diff, z));
  brem = {0.5 * brem.re, 0.5 * brem.im};
  const cplx inner = cadd(one, cdiv(cmul(d_cur, z), brem));
  const cplx rem = cmul(brem, csub(csqrt_(inner), one));
  const cplx An = cadd(A_cur, cmul(rem, A_prev));
  const cplx Bn = cadd(B_cur, cmul(rem, B_prev));
  return cdiv(An, Bn);
}

}  // namespace nlc
