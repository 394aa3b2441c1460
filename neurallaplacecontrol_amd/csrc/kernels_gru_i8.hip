// ReverseGRUEncoder (w_nl.py:14-29) with its hidden-state GEMMs on the INT8 matrix pipe (experimental, `gru_gemm = 1`; g = 64,
// wave-sized tiles -- the headline launch of K T windows).
//
// gru_encode_kernel (kernels_gru.hip) spends 0.72 of every SIMD cycle in v_mfma_f64_16x16x4_f64 and 0.24 in the FP64 gate math,
// and the two cannot overlap: an FP64 MFMA holds the SIMD's vector issue for its 64 clocks.  The operands of the hidden-state
// GEMMs are bounded (GRU states in [-1, 1], constant weights), so here they are 54-bit fixed point cut into seven signed 8-bit
// digits and multiplied digit by digit with v_mfma_i32_16x16x64_i8 (nlc_i8gemm.h: ONE instruction covers the K = 64 of a gate tile;
// 34 per tile against 16 FP64 MFMAs, a third of the matrix-pipe time, and the VALU runs beside them).  Same dataflow as the FP64
// kernel otherwise: one wavefront per 16 windows, gate tiles in the FP64 MFMA's accumulator layout, the same two-wide gate math,
// hidden states parked in LDS for the (h - n) z + n update; the B operands of the next GEMMs are the DIGITS of the new state,
// built in registers chunk by chunk as the gates produce it.  The layer-0 input GEMM (K = 4) and linear_out stay FP64 MFMAs.
// Results differ from the FP64 kernel's at the level of either path's own rounding (tools/i8gemm_check.hip: both within 5 x 2^-53
// of the row's sum of |w h| of the exact product); tests/test_gpu_i8_gemm.py holds the latents to 1e-12.
#include "nlc_device.h"
#include "nlc_gru_tile.h"
#include "nlc_i8gemm.h"
#include "nlc_kernels.h"

namespace nlc {

// pre += W_tile h for one 16-row gate tile: seven digit fragments of the tile (16 B per lane each), 34 i8 MFMAs, recombination
template <bool MERGE>
__device__ __forceinline__ v4d i8_gate(const signed char* __restrict__ tile, const i8::v4i (&dig)[i8::kDigits], const double* __restrict__ rs,
                                       int j, int q, int lane, const v4d& pre) {
  i8::v4i a[i8::kDigits], acc[i8::kLevels];
  i8::load_tile(a, tile, lane);
#pragma unroll
  for (int l = 0; l < i8::kLevels; ++l) acc[l] = i8::v4i{0, 0, 0, 0};
  i8::tile_mfma(acc, a, dig);
  return i8::recombine<MERGE>(acc, load_bias_tile(rs, j, q), pre);
}
// the same for an accumulator two GEMMs feed (layer 1's reset / update gates: W_ih h0 + W_hh h1, one row scale)
__device__ __forceinline__ v4d i8_gate2(const signed char* __restrict__ tile_a, const i8::v4i (&dig_a)[i8::kDigits], const signed char* __restrict__ tile_b,
                                        const i8::v4i (&dig_b)[i8::kDigits], bool second, const double* __restrict__ rs, int j, int q, int lane,
                                        const v4d& pre) {
  i8::v4i a[i8::kDigits], acc[i8::kLevels];
  i8::load_tile(a, tile_a, lane);
#pragma unroll
  for (int l = 0; l < i8::kLevels; ++l) acc[l] = i8::v4i{0, 0, 0, 0};
  i8::tile_mfma(acc, a, dig_a);
  if (second) {
    i8::load_tile(a, tile_b, lane);
    i8::tile_mfma(acc, a, dig_b);
  }
  return i8::recombine<false>(acc, load_bias_tile(rs, j, q), pre);
}

#ifdef NLC_I8_SAME_TILE  // tools only (timing experiment): every tile reads the first one's fragments -- L1-resident weights, wrong results
constexpr size_t kI8TileBytes = 0;
#else
constexpr size_t kI8TileBytes = (size_t)i8::kDigits * 64 * 16;  // one gate tile's digit fragments
#endif

template <int G>
__device__ __forceinline__ double gru_encode_tile_i8(const GruArgs& a, int lane, int64_t wc, int64_t kk, int tt, double* __restrict__ H0,
                                                     double* __restrict__ H1) {
  static_assert(G == 64, "one i8 MFMA covers K = 64");
  constexpr int GT = G / 16, KS = G / 4;
  const int q = lane >> 4;
  double in_mean = 0.0, in_std = 1.0;
  if (q < a.nin) {
    in_mean = a.mean[q];
    in_std = a.std[q];
  }
  const int ab_off = (a.mode == 1) ? (int)(kk / a.Kep) * a.B : 0;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    H0[ks * 64 + lane] = 0.0;
    H1[ks * 64 + lane] = 0.0;
  }
  // digits of the two layers' states (h_0 = 0: every digit is 0), dig[i][c] = digit i of the lane's entries 4 c .. 4 c + 3
  i8::v4i S0[i8::kDigits], S1[i8::kDigits];
#pragma unroll
  for (int i = 0; i < i8::kDigits; ++i) {
    S0[i] = i8::v4i{0, 0, 0, 0};
    S1[i] = i8::v4i{0, 0, 0, 0};
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
        if (q < a.nact)
          raw = (i < a.B - 1) ? a.abuf[(ab_off + 1 + i) * a.nact + q]
                              : a.u_scale * a.perturbed[(kk * a.T + (i - (a.B - 1))) * a.nact + q];
        else
          raw = (double)(a.B - 1 - j_win);
      }
      xin = (raw - in_mean) / in_std;
    } else if (q == 3) {
      xin = 1.0;
    }
    // ---------------- layer 0: input side one FP64 k-step (as gru_encode_tile), hidden side on the i8 pipe
    i8::v4i Sn[i8::kDigits];
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      gptr wp = opaque(a.Wih0p + (size_t)j * 3 * 64);
      v4d ar = mfma(wp[lane], xin, splat(0.0));
      v4d az = mfma(wp[64 + lane], xin, splat(0.0));
      const v4d ain = mfma(wp[128 + lane], xin, splat(0.0));
      v4d ahn = load_bias_tile(a.bhn0, j, q);
      if (s > 0) {
        const signed char* t = a.Whh0d + (size_t)j * 3 * kI8TileBytes;
        ar = i8_gate<true>(t, S0, a.rs_hh0, j, q, lane, ar);
        az = i8_gate<true>(t + kI8TileBytes, S0, a.rs_hh0 + G, j, q, lane, az);
        ahn = i8_gate<true>(t + 2 * kI8TileBytes, S0, a.rs_hh0 + 2 * G, j, q, lane, ahn);
      }
      const v4d hold = {H0[(4 * j + 0) * 64 + lane], H0[(4 * j + 1) * 64 + lane], H0[(4 * j + 2) * 64 + lane],
                        H0[(4 * j + 3) * 64 + lane]};
      const v4d hn = gru_gates(ar, az, ain, ahn, hold);
      // (the image is only read back for this chunk's own update: the GEMMs read the digits)
#pragma unroll
      for (int r = 0; r < 4; ++r) H0[(4 * j + r) * 64 + lane] = hn[r];
      i8::slice_chunk(Sn, j, hn);
    }
#pragma unroll
    for (int i = 0; i < i8::kDigits; ++i) S0[i] = Sn[i];
    // ---------------- layer 1
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      v4d ar = load_bias_tile(a.brz1, j, q);
      v4d az = load_bias_tile(a.brz1, GT + j, q);
      v4d ain = load_bias_tile(a.bin1, j, q);
      v4d ahn = load_bias_tile(a.bhn1, j, q);
      const signed char* ti = a.Wih1d + (size_t)j * 3 * kI8TileBytes;
      const signed char* th = a.Whh1d + (size_t)j * 3 * kI8TileBytes;
      ar = i8_gate2(ti, S0, th, S1, s > 0, a.rs_ih1, j, q, lane, ar);
      az = i8_gate2(ti + kI8TileBytes, S0, th + kI8TileBytes, S1, s > 0, a.rs_ih1 + G, j, q, lane, az);
      ain = i8_gate<true>(ti + 2 * kI8TileBytes, S0, a.rs_ih1 + 2 * G, j, q, lane, ain);
      if (s > 0) ahn = i8_gate<true>(th + 2 * kI8TileBytes, S1, a.rs_hh1 + 2 * G, j, q, lane, ahn);
      const v4d hold = {H1[(4 * j + 0) * 64 + lane], H1[(4 * j + 1) * 64 + lane], H1[(4 * j + 2) * 64 + lane],
                        H1[(4 * j + 3) * 64 + lane]};
      const v4d hn = gru_gates(ar, az, ain, ahn, hold);
#pragma unroll
      for (int r = 0; r < 4; ++r) H1[(4 * j + r) * 64 + lane] = hn[r];
      i8::slice_chunk(Sn, j, hn);
    }
#pragma unroll
    for (int i = 0; i < i8::kDigits; ++i) S1[i] = Sn[i];
  }
  // ---------------- linear_out (2 x g): rows 0,1 of one output tile, FP64
  v4d o[1];
  o[0] = splat(0.0);
  gemm_acc<1, KS>(o, a.Wop, lane, [&](int ks) { return H1[ks * 64 + lane]; });
  return o[0][0] + a.bo[q < 2 ? q : 0];
}

#ifndef NLC_I8_WAVES  // tools only: 1 = one wavefront per SIMD (512 registers, one workgroup per CU)
#define NLC_I8_WAVES 2
#endif
__global__ __launch_bounds__(256, NLC_I8_WAVES) void gru_encode_i8_kernel(const GruArgs a) {
  constexpr int G = 64, KS = G / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, c = lane & 15;
  const int64_t w = ((int64_t)blockIdx.x * 4 + wave) * 16 + c;
  const bool valid = w < a.N;
  const int64_t wc = valid ? w : a.N - 1;
  int64_t kk = 0;
  int tt = 0;
  if (a.mode == 1) {
    kk = wc / a.Tc;
    tt = a.t0 + (int)(wc - kk * a.Tc);
  }
  __shared__ double Hs[4][2][KS * 64];
#if NLC_I8_WAVES == 1
  __shared__ double pad_[4096];  // + 32 KB: a second workgroup does not fit the CU
  if (a.N < 0) pad_[threadIdx.x] = 0.0;
#endif
  const double o = gru_encode_tile_i8<G>(a, lane, wc, kk, tt, Hs[wave][0], Hs[wave][1]);
  if (valid && q < 2) {
    const int64_t wo = (a.mode == 1) ? kk * a.T + tt : w;
    a.out[wo * 2 + q] = o;
  }
}

hipError_t launch_gru_encode_i8(const GruArgs& a, hipStream_t s) {
  if (a.N <= 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.N + 63) / 64);
  hipLaunchKernelGGL(gru_encode_i8_kernel, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace nlc
