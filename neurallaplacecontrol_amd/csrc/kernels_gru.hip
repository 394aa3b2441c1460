// ReverseGRUEncoder (w_nl.py:14-29) on FP64 matrix cores.
//
// One wavefront encodes 16 action windows.  Hidden state and gate pre-activations live in MFMA
// accumulator layout (feature on rows/registers, window on columns/lanes), so the recurrence
// h -> gates -> h' never leaves registers.  Weights are the MFMA A operand, streamed fragment-packed
// from L2; the 16 windows are the B/D columns.  The encoder input does not depend on the state
// (SURVEY F6), so the planner runs this ONCE over all K*T windows instead of inside the horizon loop.
//
// Roofline: FP64 MFMA bound.  Per 16 windows (B = 4): B*MT (input) + (B-1)*KS*MT (l0 hh) + B*KS*MT (l1 ih)
// + (B-1)*KS*MT (l1 hh) + KS (head) = 1984 MFMAs of 2048 flop; HBM traffic is nu*8 B in + 16 B out per window.
#include <cstdlib>

#include "nlc_device.h"
#include "nlc_kernels.h"

namespace nlc {

// (Measured on MI355X, cfg2: gates replaced by plain FMAs (NLC_GRU_DBG=1) -> 2.65 ms vs 3.43 ms, i.e. MFMA + weight
// streaming 2.65 ms, gate transcendentals 0.8 ms; a two-k-step fragment prefetch changed nothing: loads are hidden.)
// Gate GEMMs are processed in CHUNKS of one 16-feature tile per gate (r_j, z_j, n_j): only four accumulator
// tiles are live at a time instead of sixteen, which keeps the kernel under 256 VGPRs -> two waves per SIMD,
// so one wave's gate transcendentals (FP64 VALU) overlap the other wave's MFMAs.
// Chunk-packed weights: Wc[((j*KS + ks)*3 + g)*64 + lane], g in {r, z, n}: row g*G + 16 j + (lane & 15).
template <int KS, typename BF>
__device__ __forceinline__ void chunk_gemm(v4d& c0, v4d& c1, v4d& c2, const double* __restrict__ wc, int lane, BF bfrag) {
  double a_cur[3], a_nxt[3];
  gptr p = opaque(wc);
#pragma unroll
  for (int g = 0; g < 3; ++g) a_cur[g] = p[g * 64 + lane];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) {
      p = opaque(p + 3 * 64);
#pragma unroll
      for (int g = 0; g < 3; ++g) a_nxt[g] = p[g * 64 + lane];
    }
    const double b = bfrag(ks);
    c0 = mfma(a_cur[0], b, c0);
    c1 = mfma(a_cur[1], b, c1);
    c2 = mfma(a_cur[2], b, c2);
#pragma unroll
    for (int g = 0; g < 3; ++g) a_cur[g] = a_nxt[g];
    __builtin_amdgcn_sched_barrier(0);
  }
}

// DBG = 1: timing experiment only (env NLC_GRU_DBG=1): gates replaced by a few FMAs -> MFMA + load time
template <int DBG>
__device__ __forceinline__ v4d gru_gates(const v4d& ar, const v4d& az, const v4d& ain, const v4d& ahn, const v4d& hold) {
  v4d hnew;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (DBG == 1) {
      hnew[r] = 0.25 * ar[r] + 0.125 * az[r] + 0.01 * (ain[r] + ahn[r]) + 0.5 * hold[r];
      continue;
    }
    const double rg = m::sigmoid_d(ar[r]);
    const double zg = m::sigmoid_d(az[r]);
    const double ng = m::tanh_d(ain[r] + rg * ahn[r]);
    hnew[r] = (1.0 - zg) * ng + zg * hold[r];
  }
  return hnew;
}

template <int G, int DBG>
__global__ __launch_bounds__(256, 2) void gru_encode_kernel(const GruArgs a) {
  constexpr int GT = G / 16;   // tiles per gate = chunks
  constexpr int KS = G / 4;    // k-steps over the hidden dimension
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
    kk = wc / a.Tc;
    tt = a.t0 + (int)(wc - kk * a.Tc);
  }

  v4d h0[GT], h1[GT], hn[GT];
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
    // ---------------- layer 0: input side is one k-step (K = nin padded to 4, bias folded into column 3)
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      gptr wp = opaque(a.Wih0p + (size_t)j * 3 * 64);
      v4d ar = mfma(wp[lane], xin, splat(0.0));
      v4d az = mfma(wp[64 + lane], xin, splat(0.0));
      v4d ain = mfma(wp[128 + lane], xin, splat(0.0));
      v4d ahn = load_bias_tile(a.bhn0, j, q);
      if (s > 0)
        chunk_gemm<KS>(ar, az, ahn, a.Whh0p + (size_t)j * KS * 3 * 64, lane, [&](int ks) { return h0[ks >> 2][ks & 3]; });
      hn[j] = gru_gates<DBG>(ar, az, ain, ahn, h0[j]);
    }
#pragma unroll
    for (int j = 0; j < GT; ++j) h0[j] = hn[j];
    // ---------------- layer 1
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      v4d ar = load_bias_tile(a.brz1, j, q);
      v4d az = load_bias_tile(a.brz1, GT + j, q);
      v4d ain = load_bias_tile(a.bin1, j, q);
      v4d ahn = load_bias_tile(a.bhn1, j, q);
      chunk_gemm<KS>(ar, az, ain, a.Wih1p + (size_t)j * KS * 3 * 64, lane, [&](int ks) { return h0[ks >> 2][ks & 3]; });
      if (s > 0)
        chunk_gemm<KS>(ar, az, ahn, a.Whh1p + (size_t)j * KS * 3 * 64, lane, [&](int ks) { return h1[ks >> 2][ks & 3]; });
      hn[j] = gru_gates<DBG>(ar, az, ain, ahn, h1[j]);
    }
#pragma unroll
    for (int j = 0; j < GT; ++j) h1[j] = hn[j];
  }
  // ---------------- linear_out (2 x g): rows 0,1 of one output tile
  v4d o[1];
  o[0] = splat(0.0);
  gemm_acc<1, KS>(o, a.Wop, lane, [&](int ks) { return h1[ks >> 2][ks & 3]; });
  if (valid && q < 2) {
    const int64_t wo = (a.mode == 1) ? kk * a.T + tt : w;  // (k, t) -> row of the (K, T, 2) latent tensor
    a.out[wo * 2 + q] = o[0][0] + a.bo[q];
  }
}

hipError_t launch_gru_encode(const GruArgs& a, int g, hipStream_t s) {
  if (a.N <= 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.N + 63) / 64);
  static const bool dbg = std::getenv("NLC_GRU_DBG") && std::atoi(std::getenv("NLC_GRU_DBG")) == 1;
  if (g == 64 && dbg) {
    hipLaunchKernelGGL((gru_encode_kernel<64, 1>), dim3(grid), dim3(256), 0, s, a);
  } else if (g == 64) {
    hipLaunchKernelGGL((gru_encode_kernel<64, 0>), dim3(grid), dim3(256), 0, s, a);
  } else if (g == 32) {
    hipLaunchKernelGGL((gru_encode_kernel<32, 0>), dim3(grid), dim3(256), 0, s, a);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace nlc
