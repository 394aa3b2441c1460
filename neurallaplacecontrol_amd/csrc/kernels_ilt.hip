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
#include "nlc_device.h"
#include "nlc_kernels.h"

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
      const double t = a.t_batched ? a.t[row] : a.t[j];
      const double Tt = a.scale * t;
      const double gamma = a.alpha - a.log_tol / (a.scale * Tt);
      const int k = col < a.S ? col : col - a.S;
      const double im = kPi * (double)k / Tt;
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

// ------------------------------------------------------------------ Fourier series
// x[n,c] = e^{gamma t}/T * sum_k w_k Re(F_k e^{i pi k/scale}),  w_0 = 1/2, t/T = 1/scale for every t.
// Rows (n,c) are contiguous runs of S doubles in theta/phi.  A block streams ROWS rows with perfectly
// coalesced loads (lane i <-> flat element i), parks the per-element contribution in LDS and lets one
// thread per row add its S terms (row stride padded odd: conflict-free ds_read_b64).

__global__ __launch_bounds__(256) void ilt_fourier_kernel(const IltArgs a) {
  extern __shared__ double lds[];
  const int S = a.S;
  const int SP = S | 1;
  double* cw = lds;              // [S]  w_k cos(pi k/scale)
  double* sw = lds + S;          // [S] -w_k sin(pi k/scale)
  double* val = lds + 2 * S;     // [rows][SP]
  const int kIltRows = a.rows;
  for (int k = threadIdx.x; k < S; k += blockDim.x) {
    const double w = k == 0 ? 0.5 : 1.0;
    double sn, cs;
    if (a.scale == 2.0) {  // exact powers of i
      const int r = k & 3;
      cs = (r == 0) ? 1.0 : (r == 2 ? -1.0 : 0.0);
      sn = (r == 1) ? 1.0 : (r == 3 ? -1.0 : 0.0);
    } else {
      const double ang = kPi * (double)k / a.scale;
      sn = sin(ang);
      cs = cos(ang);
    }
    cw[k] = w * cs;
    sw[k] = -w * sn;
  }
  __syncthreads();
  const int64_t rows_total = a.N * a.d;
  const int64_t nblk = (rows_total + kIltRows - 1) / kIltRows;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t row0 = blk * kIltRows;
    const int64_t rows_here = (rows_total - row0 < kIltRows) ? (rows_total - row0) : kIltRows;
    const int64_t elems = rows_here * S;
    const int64_t base = row0 * S;
    // (r, k) of this thread's first element, then advanced by 256 elements per iteration.  Loads are issued
    // in batches of UB iterations before any arithmetic so each lane keeps 2*UB 8-byte loads in flight
    // (one load pair per iteration leaves the HBM pipe latency-bound at ~3 TB/s).
    int r = threadIdx.x / S, k = threadIdx.x - r * S;
    const int dr = 256 / S, dk = 256 - dr * S;
    constexpr int UB = 6;
    for (int64_t e0 = threadIdx.x; e0 < elems; e0 += 256 * UB) {
      double th[UB], ph[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int64_t e = e0 + 256 * u;
        th[u] = 0.0;
        ph[u] = 0.0;
        if (e < elems) {
          th[u] = __builtin_nontemporal_load(a.theta + base + e);
          ph[u] = __builtin_nontemporal_load(a.phi + base + e);
        }
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int64_t e = e0 + 256 * u;
        if (e < elems) {
          const double rad = m::tan_0_halfpi(ph[u] / 2.0 + kPi / 4.0);
          double sn, cs;
          m::sincos_bounded(th[u], &sn, &cs);
          val[r * SP + k] = rad * (cw[k] * cs + sw[k] * sn);
          r += dr;
          k += dk;
          if (k >= S) {
            k -= S;
            r += 1;
          }
        }
      }
    }
    __syncthreads();
    if ((int64_t)threadIdx.x < rows_here) {
      const double* v = val + threadIdx.x * SP;
      double acc = 0.0;
      for (int kk = 0; kk < S; ++kk) acc += v[kk];
      const int64_t row = row0 + threadIdx.x;
      const double t = a.t[row / a.d];
      const double Tt = a.scale * t;
      const double gamma = a.alpha - a.log_tol / (a.scale * Tt);
      a.x[row] = exp(gamma * t) / Tt * acc;
    }
    __syncthreads();
  }
}

hipError_t launch_ilt_fourier(const IltArgs& a_in, hipStream_t s) {
  IltArgs a = a_in;
  const int64_t rows_total = a.N * a.d;
  if (rows_total <= 0) return hipSuccess;
  // rows per block: one thread per row for the final sum, LDS tile kept under the 64 KiB default
  const int SP = a.S | 1;
  int rows = (int)((60 * 1024 / 8 - 2 * a.S) / SP);
  if (rows > 256) rows = 256;
  if (rows < 1) return hipErrorInvalidValue;
  a.rows = rows;
  const int64_t nblk = (rows_total + rows - 1) / rows;
  const unsigned grid = (unsigned)(nblk < 2048 ? nblk : 2048);
  const size_t shmem = (size_t)(2 * a.S + rows * SP) * sizeof(double);
  hipLaunchKernelGGL(ilt_fourier_kernel, dim3(grid), dim3(256), shmem, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ de Hoog, Knight & Stokes
struct cplx {
  double re, im;
};
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ cplx cdiv(cplx a, cplx b) {
  const double den = b.re * b.re + b.im * b.im;
  return {(a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den};
}
__device__ __forceinline__ cplx csqrt_(cplx z) {
  // principal branch
  const double mag = hypot(z.re, z.im);
  double re = sqrt(0.5 * (mag + fabs(z.re)));
  double im = (re == 0.0) ? 0.0 : 0.5 * z.im / re;
  if (z.re < 0.0) {
    const double t = re;
    re = fabs(im);
    im = copysign(t, z.im);
  }
  return {re, im};
}

// One thread per (point, dim) row.  F_k is staged through LDS by the whole block with coalesced loads; the
// QD columns q[0..2M), e[0..2M] live in registers (compile-time M, statically indexed, in-place rhombus
// updates), and the continued-fraction recurrence consumes d_i as soon as a column produces it.
template <int M>
__global__ __launch_bounds__(64) void ilt_dehoog_kernel(const IltArgs a) {
  constexpr int S = 2 * M + 1;
  constexpr int SP = S | 1;
  constexpr int ROWS = 64;
  __shared__ double fr[ROWS * SP];
  __shared__ double fi[ROWS * SP];
  const int64_t rows_total = a.N * a.d;
  const int64_t nblk = (rows_total + ROWS - 1) / ROWS;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t row0 = blk * ROWS;
    const int64_t rows_here = (rows_total - row0 < ROWS) ? (rows_total - row0) : ROWS;
    const int64_t elems = rows_here * S;
    const int64_t base = row0 * S;
    int r = threadIdx.x / S, k = threadIdx.x - r * S;
    constexpr int dr = ROWS / S, dk = ROWS - dr * S;
    for (int64_t e = threadIdx.x; e < elems; e += ROWS) {
      const double theta = a.theta[base + e];
      const double phi = a.phi[base + e];
      const double rad = m::tan_0_halfpi(phi / 2.0 + kPi / 4.0);
      double sn, cs;
      m::sincos_bounded(theta, &sn, &cs);
      fr[r * SP + k] = rad * cs;
      fi[r * SP + k] = rad * sn;
      r += dr;
      k += dk;
      if (k >= S) {
        k -= S;
        r += 1;
      }
    }
    __syncthreads();
    if ((int64_t)threadIdx.x < rows_here) {
      const double* pr = fr + threadIdx.x * SP;
      const double* pi = fi + threadIdx.x * SP;
      cplx q[2 * M], e[2 * M + 1];
      const cplx f0 = {pr[0], pi[0]};
      const cplx d0 = {0.5 * f0.re, 0.5 * f0.im};
      {
        cplx prev = f0;
#pragma unroll
        for (int i = 0; i < 2 * M; ++i) {
          const cplx cur = {pr[i + 1], pi[i + 1]};
          q[i] = cdiv(cur, i == 0 ? d0 : prev);
          prev = cur;
        }
      }
#pragma unroll
      for (int i = 0; i <= 2 * M; ++i) e[i] = {0.0, 0.0};
      const int64_t row = row0 + threadIdx.x;
      const double t = a.t[row / a.d];
      const double Tt = a.scale * t;
      const double gamma = a.alpha - a.log_tol / (a.scale * Tt);
      const double ang = kPi * (t / Tt);
      const cplx z = {cos(ang), sin(ang)};
      // A/B continued-fraction recurrence, fed with d_1, d_2, ... as they appear
      cplx A_prev = {0.0, 0.0}, A_cur = d0, B_prev = {1.0, 0.0}, B_cur = {1.0, 0.0};
      cplx d_last = {0.0, 0.0}, d_cur = {0.0, 0.0};
      auto feed = [&](cplx d, bool advance) {
        d_last = d_cur;
        d_cur = d;
        if (advance) {
          const cplx dz = cmul(d, z);
          const cplx An = cadd(A_cur, cmul(dz, A_prev));
          const cplx Bn = cadd(B_cur, cmul(dz, B_prev));
          A_prev = A_cur;
          A_cur = An;
          B_prev = B_cur;
          B_cur = Bn;
        }
      };
      // d_1 = -q[0,0]
      feed({-q[0].re, -q[0].im}, true);
#pragma unroll
      for (int rr = 1; rr <= M; ++rr) {
        const int mr = 2 * (M - rr) + 1;
        // e column rr from q column rr-1 and e column rr-1 (in place, ascending i)
#pragma unroll
        for (int i = 0; i < 2 * M; ++i) {
          if (i < mr) e[i] = cadd(csub(q[i + 1 < 2 * M ? i + 1 : i], q[i]), e[i + 1]);
        }
        // d_{2 rr} = -e[0, rr]   (the last one, d_{2M}, only enters the remainder)
        feed({-e[0].re, -e[0].im}, rr != M);
        if (rr != M) {
          const int mrq = 2 * (M - rr - 1) + 1 + 2;
#pragma unroll
          for (int i = 0; i < 2 * M - 1; ++i) {
            if (i < mrq) q[i] = cdiv(cmul(q[i + 1], e[i + 1]), e[i]);
          }
          // d_{2 rr + 1} = -q[0, rr]
          feed({-q[0].re, -q[0].im}, true);
        }
      }
      // here d_last = d_{2M-1}, d_cur = d_{2M}; recurrence has run for i = 1 .. 2M-1
      const cplx diff = csub(d_last, d_cur);
      const cplx one = {1.0, 0.0};
      cplx brem = cadd(one, cmul(diff, z));
      brem = {0.5 * brem.re, 0.5 * brem.im};
      const cplx inner = cadd(one, cdiv(cmul(d_cur, z), brem));
      const cplx rem = cmul(brem, csub(csqrt_(inner), one));
      const cplx An = cadd(A_cur, cmul(rem, A_prev));
      const cplx Bn = cadd(B_cur, cmul(rem, B_prev));
      const cplx res = cdiv(An, Bn);
      a.x[row] = exp(gamma * t) / Tt * res.re;
    }
    __syncthreads();
  }
}

hipError_t launch_ilt_dehoog(const IltArgs& a, hipStream_t s) {
  const int64_t rows_total = a.N * a.d;
  if (rows_total <= 0) return hipSuccess;
  const int64_t nblk = (rows_total + 63) / 64;
  const unsigned grid = (unsigned)(nblk < 8192 ? nblk : 8192);
  switch (a.S) {
    case 33:
      hipLaunchKernelGGL(ilt_dehoog_kernel<16>, dim3(grid), dim3(64), 0, s, a);
      break;
    case 17:
      hipLaunchKernelGGL(ilt_dehoog_kernel<8>, dim3(grid), dim3(64), 0, s, a);
      break;
    case 9:
      hipLaunchKernelGGL(ilt_dehoog_kernel<4>, dim3(grid), dim3(64), 0, s, a);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace nlc
