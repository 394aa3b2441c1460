// ReverseGRUEncoder (w_nl.py:14-29) on FP64 matrix cores.
//
// One wavefront encodes 16 action windows.  Hidden state and gate pre-activations live in MFMA
// accumulator layout (feature on rows/registers, window on columns/lanes), so the recurrence
// h -> gates -> h' never leaves registers.  Weights are the MFMA A operand, streamed fragment-packed
// from L2; the 16 windows are the B/D columns.  The encoder input does not depend on the state
// (SURVEY F6), so the planner runs this ONCE over all K*T windows instead of inside the horizon loop.
//
// Roofline: FP64 MFMA bound.  Per 16 windows: 4*MT (input) + 3*KS*MT (l0 hh) + 4*KS*MT (l1 ih)
// + 3*KS*MT (l1 hh) + KS (head) MFMAs of 2048 flop; HBM traffic is nu*8 B in + 16 B out per window.
#include "nlc_device.h"
#include "nlc_kernels.h"

namespace nlc {

// Hidden-side GEMM of one GRU step: r,z rows of W_hh accumulate into acc (joining the input side),
// n rows into accn so that r * (W_hn h + b_hn) can be formed afterwards.
template <int GT, int KS>
__device__ __forceinline__ void gru_hidden_gemm(v4d (&acc)[3 * GT], v4d (&accn)[GT], const double* __restrict__ whp,
                                                int lane, const v4d (&h)[GT]) {
  constexpr int MT = 3 * GT;
  double a_cur[MT], a_nxt[MT];
  gptr p = opaque(whp);
#pragma unroll
  for (int m = 0; m < MT; ++m) a_cur[m] = p[m * 64 + lane];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) {
      p = opaque(p + MT * 64);
#pragma unroll
      for (int m = 0; m < MT; ++m) a_nxt[m] = p[m * 64 + lane];
    }
    const double b = h[ks >> 2][ks & 3];
#pragma unroll
    for (int m = 0; m < 2 * GT; ++m) acc[m] = mfma(a_cur[m], b, acc[m]);
#pragma unroll
    for (int m = 0; m < GT; ++m) accn[m] = mfma(a_cur[2 * GT + m], b, accn[m]);
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int GT>
__device__ __forceinline__ void gru_gates(const v4d (&acc)[3 * GT], const v4d (&accn)[GT], v4d (&h)[GT]) {
#pragma unroll
  for (int j = 0; j < GT; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double rg = m::sigmoid_d(acc[j][r]);
      const double zg = m::sigmoid_d(acc[GT + j][r]);
      const double ng = m::tanh_d(acc[2 * GT + j][r] + rg * accn[j][r]);
      h[j][r] = (1.0 - zg) * ng + zg * h[j][r];
    }
  }
}

template <int G>
__global__ __launch_bounds__(256) void gru_encode_kernel(const GruArgs a) {
  constexpr int GT = G / 16;   // tiles per gate
  constexpr int KS = G / 4;    // k-steps over the hidden dimension
  constexpr int MT = 3 * GT;   // gate tiles r|z|n
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, c = lane & 15;
  const int64_t w = ((int64_t)blockIdx.x * 4 + wave) * 16 + c;
  const bool valid = w < a.N;
  const int64_t wc = valid ? w : a.N - 1;

  // per-lane input normalisation constants (lane q feeds input dim q; q == 3 feeds the bias column)
  double in_mean = 0.0, in_std = 1.0;
  if (q < a.nin) {
    in_mean = a.mean[q];
    in_std = a.std[q];
  }
  int64_t kk = 0;
  int tt = 0;
  if (a.mode == 1) {
    kk = wc / a.T;
    tt = (int)(wc - kk * a.T);
  }

  v4d h0[GT], h1[GT];
#pragma unroll
  for (int j = 0; j < GT; ++j) {
    h0[j] = splat(0.0);
    h1[j] = splat(0.0);
  }

  for (int s = 0; s < a.B; ++s) {
    // reversed time: GRU step s consumes window element B-1-s  (torch.flip, w_nl.py:27)
    const int j_win = a.B - 1 - s;
    double xin = 0.0;
    if (q < a.nin) {
      double raw;
      if (a.mode == 0) {
        raw = a.window[(wc * a.B + j_win) * a.nin + q];
      } else {
        const int i = tt + j_win;
        raw = (i < a.B - 1) ? a.abuf[(1 + i) * a.nin + q]
                            : a.u_scale * a.perturbed[(kk * a.T + (i - (a.B - 1))) * a.nin + q];
      }
      xin = (raw - in_mean) / in_std;
    } else if (q == 3) {
      xin = 1.0;  // bias column of the packed W_ih0
    }

    v4d acc[MT];   // r: 0..GT-1, z: GT..2GT-1, n(input part): 2GT..3GT-1
    v4d accn[GT];  // n(hidden part): W_hn h + b_hn
    // ---------------- layer 0
    {
      gptr wp = opaque(a.Wih0p);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = mfma(wp[m * 64 + lane], xin, splat(0.0));
#pragma unroll
      for (int j = 0; j < GT; ++j) accn[j] = load_bias_tile(a.bhn0, j, q);
      if (s > 0) gru_hidden_gemm<GT, KS>(acc, accn, a.Whh0p, lane, h0);
      gru_gates<GT>(acc, accn, h0);
    }
    // ---------------- layer 1
    {
#pragma unroll
      for (int j = 0; j < 2 * GT; ++j) acc[j] = load_bias_tile(a.brz1, j, q);
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        acc[2 * GT + j] = load_bias_tile(a.bin1, j, q);
        accn[j] = load_bias_tile(a.bhn1, j, q);
      }
      gemm_acc<MT, KS>(acc, a.Wih1p, lane, [&](int ks) { return h0[ks >> 2][ks & 3]; });
      if (s > 0) gru_hidden_gemm<GT, KS>(acc, accn, a.Whh1p, lane, h1);
      gru_gates<GT>(acc, accn, h1);
    }
  }
  // ---------------- linear_out (2 x g): rows 0,1 of one output tile
  v4d o[1];
  o[0] = splat(0.0);
  gemm_acc<1, KS>(o, a.Wop, lane, [&](int ks) { return h1[ks >> 2][ks & 3]; });
  if (valid && q < 2) a.out[w * 2 + q] = o[0][0] + a.bo[q];
}

hipError_t launch_gru_encode(const GruArgs& a, int g, hipStream_t s) {
  if (a.N <= 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.N + 63) / 64);
  if (g == 64) {
    hipLaunchKernelGGL(gru_encode_kernel<64>, dim3(grid), dim3(256), 0, s, a);
  } else if (g == 32) {
    hipLaunchKernelGGL(gru_encode_kernel<32>, dim3(grid), dim3(256), 0, s, a);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace nlc
