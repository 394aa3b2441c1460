// de Hoog, Knight & Stokes accelerated inverse Laplace transform (BASELINE configs[4]): the second ilt_algorithm of
// torchlaplace.laplace_reconstruct (external; call site w_nl.py:137-144), restated from mpmath 1.3.0
// calculus/inverselaplace.py:476-531 (parity unpinned vs upstream, see oracle/ilt.py).  Any odd number of terms
// S = 2M + 1 <= 33.  A translation unit of its own: 16 term counts x 3 input layouts of a fully unrolled kernel.
#include "nlc_cplx.h"
#include "nlc_dehoog_row.h"
#include "nlc_device.h"
#include "nlc_kernels.h"

namespace nlc {

// ------------------------------------------------------------------ de Hoog, Knight & Stokes
// One thread per (point, dim) row; a block is one wavefront = 64 rows.
//
// The QD table is built anti-diagonal by anti-diagonal ("progressive" form): Laplace term a_n extends every column
// by one entry, and the new diagonal is computed in place over the previous one -- D[c-1] holds the entry of column c
// (c = 1: q_1, 2: e_1, 3: q_2, ...) on the current diagonal -- with the same rhombus rules as mpmath's column-wise
// sweep (calculus/inverselaplace.py:476-531), so every entry is the same arithmetic on the same operands:
//     e_r^(i)     = q_r^(i+1) - q_r^(i) + e_(r-1)^(i+1)            (even column)
//     q_(r+1)^(i) = q_r^(i+1) e_r^(i+1) / e_r^(i)                  (odd column)
// The last entry of diagonal n is the continued-fraction coefficient d_n = -(entry at i = 0), consumed at once by the
// A/B recurrence.  Storage is ONE diagonal (2M complex = 128 VGPRs at M = 16) instead of the two full columns of the
// column-wise form (354 VGPRs, one wave per SIMD), and the F_k rows are staged through LDS in chunks of CH terms, so
// two waves per SIMD share the VALU (a single wave issues FP64 VALU at half rate: tools/ubench_valu64.hip).
// Measured (N = 655 360 points, d = 5): S = 33 2.50 -> 1.39 ms, S = 17 0.70 -> 0.48 ms; 9.6 k VALU instructions per
// 64 rows at S = 33 (7.3 k QD + 2.3 k sphere->complex conversion): the kernel is bound by its FP64 instruction
// count.  An intermediate version (1.65 ms) looked latency-bound at 66 % VALU utilisation; the PMC traffic counters
// showed the real cause -- the compiler had deferred the B recurrence to the end of the kernel and spilled every
// d_n z to scratch, 1 GB of HBM writes and 3.5x the algorithmic traffic per launch (profiles/r1g_pmc_dehoog.json
// vs r1h_pmc_kernels.json); the per-diagonal fence below now pins both recurrences.
// FMODE 0: (theta, phi) rows, sphere -> complex conversion here; 1: F (re, im) rows (N, d, S); both staged through LDS.
// FMODE 2 (planner path): F is SLOT-major (8*nt3, N) as the representation kernel's MFMA epilogue stores it; a wavefront
// owns 64 consecutive samples of ONE dim, so term k of its rows is one full 512-B line -- no LDS, no barrier, and the
// kernel reads exactly the bytes it needs (row-major rows of 33 doubles straddle the 17-term chunks: 1.8x the traffic).
// Round 3, the 1.8x re-read of the row-major modes -- two single-read forms were built and measured on
// N = 655 360, d = 5, S = 33 (1.31 ms, 3.17 GB at the fabric counter as shipped): staging by ROW GROUPS (the flat 32-row half
// of the block read once, coalesced, through the same 17 KB of LDS) with (a) all S terms of a lane's row pulled into
// registers: 256 VGPRs, 104 spilled, 1.54 ms and 0.7 GB of scratch WRITES per launch; (b) only the high terms M + 1 .. 2M in
// registers and the low ones back in LDS: 168 spilled.  The pending terms and the growing QD diagonal do fit one budget on
// paper (at most 2M complex together), but not in the allocator's hands, and a full LDS image (34 KB per wavefront) means
// one wave per SIMD, i.e. the FP64 VALU -- the kernel's actual bound -- at half rate.  The chunked form stays; the planner
// path (FMODE 2) reads exactly its bytes.
// Term source of the row-major modes: the block's 64 rows are staged through LDS in chunks of CH terms, coalesced over
// (row, term) pairs; FDIRECT: F (re, im) rows, else (theta, phi) rows with the sphere -> complex conversion here.
template <int CH, bool FDIRECT>
struct DehoogLdsTerms {
  static constexpr int CP = CH | 1, ROWS = 64;
  const IltArgs& a;
  double* fr;
  double* fi;
  int lane, rows_here;
  int64_t row0;
  template <int S>
  __device__ __forceinline__ void stage(int n) {
    const int nt = (S - n < CH) ? (S - n) : CH;
    if (n != 0) __syncthreads();
#pragma nounroll
    for (int e = lane; e < ROWS * nt; e += ROWS) {
      const int r = e / nt, k = e - r * nt;
      if (r < rows_here) {
        const int64_t gi = (row0 + r) * S + n + k;
        if constexpr (FDIRECT) {  // F_k supplied directly (staged planner / model path)
          fr[r * CP + k] = a.fre[gi];
          fi[r * CP + k] = a.fim[gi];
        } else {
          const double theta = a.theta[gi];
          const double phi = a.phi[gi];
          const double rad = m::tan_0_halfpi(phi / 2.0 + kPi / 4.0);
          double sn, cs;
          m::sincos_bounded(theta, &sn, &cs);
          fr[r * CP + k] = rad * cs;
          fi[r * CP + k] = rad * sn;
        }
      }
    }
    __syncthreads();
  }
  __device__ __forceinline__ cplx term(int n) const { return {fr[lane * CP + n % CH], fi[lane * CP + n % CH]}; }
};

template <int M, int CH, int W, int FMODE>
__global__ __launch_bounds__(64, W) void ilt_dehoog_kernel(const IltArgs a) {
  constexpr int S = 2 * M + 1;
  constexpr int CP = CH | 1;
  constexpr int ROWS = 64;
  constexpr bool FDIRECT = FMODE == 1;
  constexpr bool SLOT = FMODE == 2;
  __shared__ double fr[SLOT ? 1 : ROWS * CP];
  __shared__ double fi[SLOT ? 1 : ROWS * CP];
  const int lane = threadIdx.x;
  const int64_t rows_total = a.N * a.d;
  const int64_t nsb = (a.N + ROWS - 1) / ROWS;  // SLOT: sample blocks
  const int64_t nblk = SLOT ? nsb * a.d : (rows_total + ROWS - 1) / ROWS;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    // SLOT: block = (sample block, dim); otherwise 64 consecutive (point, dim) rows
    const int cdim = SLOT ? (int)(blk % a.d) : 0;
    const int64_t n0 = SLOT ? (blk / a.d) * ROWS : 0;
    const int64_t row0 = SLOT ? 0 : blk * ROWS;
    const int rows_here = SLOT ? (int)((a.N - n0 < ROWS) ? (a.N - n0) : ROWS)
                               : (int)((rows_total - row0 < ROWS) ? (rows_total - row0) : ROWS);
    const bool valid = lane < rows_here;
    const int64_t nsmp = n0 + (valid ? lane : 0);                     // SLOT: this lane's sample
    const int64_t row = SLOT ? nsmp * a.d + cdim : row0 + (valid ? lane : 0);
    const double t = (a.t_stride ? a.t[row / a.d] : a.t[0]) / a.t_div;
    const double Tt = a.scale * t;
    const double gamma = a.alpha - a.log_tol / (a.scale * Tt);
    const double ang = kPi * (t / Tt);
    const cplx z = {cos(ang), sin(ang)};
    cplx res;
    if constexpr (SLOT) {
      // wave-uniform slot of every term; one coalesced line per term and array
      DehoogSlotTerms<CH> src{a.fre, a.fim, a.eidx + cdim * S, a.N, nsmp, {}};
      res = dehoog_row<M, CH, kDehoogSkewAlone>(src, z);  // the planner's launch: one wavefront per SIMD at most
    } else {
      DehoogLdsTerms<CH, FDIRECT> src{a, fr, fi, lane, rows_here, row0};
      res = dehoog_row<M, CH>(src, z);
    }
    if (valid) a.x[row] = exp(gamma * t) / Tt * res.re;
    if (!SLOT) __syncthreads();
  }
}

hipError_t launch_ilt_dehoog(const IltArgs& a, hipStream_t s) {
  const int64_t rows_total = a.N * a.d;
  if (rows_total <= 0) return hipSuccess;
  const bool slot = a.eidx != nullptr;
  if (slot && a.fre == nullptr) return hipErrorInvalidValue;
  if (a.S < 3 || a.S > 33 || (a.S & 1) == 0) return hipErrorInvalidValue;  // S = 2M + 1 terms, M = 1 .. 16
  const int64_t nblk = slot ? (a.N + 63) / 64 * a.d : (rows_total + 63) / 64;
  const unsigned grid = (unsigned)(nblk < 16384 ? nblk : 16384);
  // chunk length CH / waves per SIMD W: M = 16 runs at 212-249 VGPRs with two 17-term chunks (17 KB of LDS, 8 waves per
  // CU); M <= 8 at <= 160 VGPRs with 9-term chunks (9 KB, 12 waves per CU)
#define NLC_DH(MM)                                                                                               \
  case 2 * MM + 1: {                                                                                             \
    constexpr int CH = MM > 8 ? MM + 1 : 9, W = MM > 8 ? 2 : 3;                                                  \
    if (slot)                                                                                                    \
      hipLaunchKernelGGL((ilt_dehoog_kernel<MM, CH, W, 2>), dim3(grid), dim3(64), 0, s, a);                      \
    else if (a.fre != nullptr)                                                                                   \
      hipLaunchKernelGGL((ilt_dehoog_kernel<MM, CH, W, 1>), dim3(grid), dim3(64), 0, s, a);                      \
    else                                                                                                         \
      hipLaunchKernelGGL((ilt_dehoog_kernel<MM, CH, W, 0>), dim3(grid), dim3(64), 0, s, a);                      \
  } break;
  switch (a.S) {
    NLC_DH(1) NLC_DH(2) NLC_DH(3) NLC_DH(4) NLC_DH(5) NLC_DH(6) NLC_DH(7) NLC_DH(8)
    NLC_DH(9) NLC_DH(10) NLC_DH(11) NLC_DH(12) NLC_DH(13) NLC_DH(14) NLC_DH(15) NLC_DH(16)
    default:
      return hipErrorInvalidValue;
  }
#undef NLC_DH
  return hipGetLastError();
}

}  // namespace nlc
