// One 16-window tile of the ReverseGRUEncoder (w_nl.py:14-29) on FP64 matrix cores: chunked gate GEMMs, two-wide
// gate math and the tile body shared by gru_encode_kernel (kernels_gru.hip) and the fused planner kernel
// (kernels_fused.hip).  See kernels_gru.hip for the dataflow and the roofline.
#pragma once
#include "nlc_device.h"
#include "nlc_kernels.h"

namespace nlc {

// (Measured on MI355X, cfg2, round 1: gates replaced by plain FMAs -> 2.65 ms vs 3.43 ms, i.e. MFMA + weight
// streaming 2.65 ms, gate transcendentals 0.8 ms; a two-k-step fragment prefetch changed nothing: loads are hidden.)
// Gate GEMMs are processed in CHUNKS of one 16-feature tile per gate (r_j, z_j, n_j): only four accumulator
// tiles are live at a time instead of sixteen, which keeps the kernel under 256 VGPRs -> two waves per SIMD,
// so one wave's gate transcendentals (FP64 VALU) overlap the other wave's MFMAs.  (With two waves per SIMD the k-step's loads
// stay in front of its MFMAs: the MFMA / load interleave that pays at one wave per SIMD -- gemm_kstep_order -- cost this kernel
// 2.5 %, 2.967 -> 3.042 ms; round 4.)
// Chunk-packed weights: Wc[((j*KS + ks)*3 + g)*64 + lane], g in {r, z, n}: row g*G + 16 j + (lane & 15).
template <int KS>
__device__ __forceinline__ void chunk_gemm(v4d& c0, v4d& c1, v4d& c2, const double* __restrict__ wc, int lane,
                                           const double* __restrict__ hb) {
  // hb: this wave's hidden-state image in LDS, hb[ks*64 + lane] = B fragment of k-step ks (written by the same lane)
  double a_cur[3], a_nxt[3];
  gptr p = opaque(wc);
#pragma unroll
  for (int g = 0; g < 3; ++g) a_cur[g] = p[g * 64 + lane];
  double b_cur = hb[lane], b_nxt = 0.0;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) {
      p = opaque(p + 3 * 64);
#pragma unroll
      for (int g = 0; g < 3; ++g) a_nxt[g] = p[g * 64 + lane];
      b_nxt = hb[(ks + 1) * 64 + lane];
    }
    c0 = mfma(a_cur[0], b_cur, c0);
    c1 = mfma(a_cur[1], b_cur, c1);
    c2 = mfma(a_cur[2], b_cur, c2);
#pragma unroll
    for (int g = 0; g < 3; ++g) a_cur[g] = a_nxt[g];
    b_cur = b_nxt;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- two-wide gate math: the r/z sigmoids and the n tanh of TWO hidden units are evaluated in lockstep (clang
// ext-vector arithmetic = two independent FP64 instruction streams), so each wave issues two dependent chains
// instead of one and the FP64 VALU latency is covered without relying on the partner wave.
typedef double v2d __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2d fma2(v2d a, v2d b, v2d c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2d splat2(double x) { return v2d{x, x}; }
// ---- exponentials of the gate math (instruction count, the GRU kernel's FP64 VALU adds to its MFMA time)
// e^r = 1 + r + r^2 q(r) on |r| <= ln2/2 with q of degree 7 (Chebyshev interpolation of (e^r - 1 - r)/r^2: relative error
// 7.4e-14 -- the gates feed a tanh / a convex combination, the parity bar is 1e-5 and the tests hold 1e-9), evaluated as one
// Horner chain of nine FMAs ((q r + 1) r + 1).  The reduction uses ONE ln2 constant (|n| ln2 2^-53 <= 3e-14 for every
// argument that matters), the integer is the low word of the 1.5 * 2^52-shifted sum (no v_rndne / v_cvt).
__device__ __forceinline__ v2d exp_poly2(v2d r) {
  v2d q = splat2(0x1.72ad458027fbcp-19);
  q = fma2(q, r, splat2(0x1.a136bf03ec612p-16));
  q = fma2(q, r, splat2(0x1.a019c36bc053cp-13));
  q = fma2(q, r, splat2(0x1.6c166bde96885p-10));
  q = fma2(q, r, splat2(0x1.111111170bc08p-7));
  q = fma2(q, r, splat2(0x1.55555565c7e0ep-5));
  q = fma2(q, r, splat2(0x1.5555555554f96p-3));
  q = fma2(q, r, splat2(0x1.fffffffffe062p-2));
  q = fma2(q, r, splat2(1.0));
  return fma2(q, r, splat2(1.0));
}
__device__ __forceinline__ v2d exp2n2(v2d y) {  // e^y, y <= 170; flushes to 0 below ~-745
  const v2d sh = fma2(y, splat2(1.44269504088896338700e+00), splat2(6755399441055744.0));
  const v2d fn = sh - splat2(6755399441055744.0);
  const v2d r = fma2(-fn, splat2(6.93147180559945286227e-01), y);
  const v2d p = exp_poly2(r);
  return v2d{ldexp(p.x, __double2loint(sh.x)), ldexp(p.y, __double2loint(sh.y))};
}
// (v_min / v_max spelled out: through the builtins the compiler first canonicalises every operand with a v_max_f64 x, x, x
// of its own -- 96 extra FP64 instructions per GRU step and lane)
__device__ __forceinline__ double min_neg(double x, double hi) {  // min(-x, hi)
  double r;
  asm("v_min_f64 %0, -%1, %2" : "=v"(r) : "v"(x), "s"(hi));
  return r;
}
__device__ __forceinline__ double max_raw(double x, double lo) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "s"(lo));
  return r;
}
__device__ __forceinline__ double rcp_refined1(double d) {
  // v_rcp_f64 (~2^-24) + one cubic step r (1 + e + e^2), e = 1 - d r: 3 FMAs, <= 1 ulp (tools/rcp_probe.hip)
  const double r = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, r, 1.0);
  return fma(r, fma(e, e, e), r);
}
// 1 + e^{-x} for either sign (e^{-x} >= 0, nothing cancels).  The argument is clamped from above at 170 so that the
// product of FOUR such terms stays finite (beyond it the true sigmoid is < 1e-73 and this returns ~1e-74); no clamp from
// below: for very negative arguments v_ldexp_f64 flushes the term to 0 and the result is exactly 1.
__device__ __forceinline__ v2d one_plus_exp_neg2(v2d x) {
  return splat2(1.0) + exp2n2(v2d{min_neg(x.x, 170.0), min_neg(x.y, 170.0)});
}
// the reset and update gates of TWO hidden units share ONE reciprocal: with A = 1 + e^{-a}, B = 1 + e^{-b} per unit and
// R = 1 / (A0 B0 A1 B1), sigmoid(a_i) = B_i / (A_i B_i) = B_i (A_j B_j) R  (v_rcp_f64 is quarter rate: one + its three
// refinement FMAs replaced by three multiplies)
__device__ __forceinline__ void sigmoid_pair2(v2d a, v2d b, v2d* sa, v2d* sb) {
  const v2d A = one_plus_exp_neg2(a), B = one_plus_exp_neg2(b);
  const v2d P = A * B;
  const double R = rcp_refined1(P.x * P.y);
  const v2d inv = v2d{P.y, P.x} * splat2(R);  // (1 / P.x, 1 / P.y)
  *sa = B * inv;
  *sb = A * inv;
}
// tanh of two values, one reciprocal: t_i = (1 - e_i) / (1 + e_i), e = e^{-2|x|} in (0, 1], denominators in [1, 2].
// (1 - e is exact near 0 up to e's own rounding: absolute error ~1e-16, which is what a gate needs.)
__device__ __forceinline__ v2d tanh2(v2d x) {
  const v2d y2 = -2.0 * __builtin_elementwise_abs(x);
  const v2d e = exp2n2(v2d{max_raw(y2.x, -745.0), max_raw(y2.y, -745.0)});
  const v2d num = splat2(1.0) - e, d = splat2(1.0) + e;
  const double R = rcp_refined1(d.x * d.y);
  const v2d t = (num * v2d{d.y, d.x}) * splat2(R);
  return __builtin_elementwise_copysign(t, x);
}

// the window is column lane & 15 of the tile; its input dims live in the four lanes q = lane >> 4
__device__ __forceinline__ double nan_if_bad_window(bool bad_input, double value) {
  int bad = bad_input ? 1 : 0;
  bad |= __shfl_xor(bad, 16);
  bad |= __shfl_xor(bad, 32);
  return bad ? __builtin_nan("") : value;
}

__device__ __forceinline__ v4d gru_gates(const v4d& ar, const v4d& az, const v4d& ain, const v4d& ahn, const v4d& hold) {
  v4d hnew;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const v2d r2 = half ? ar.zw : ar.xy, z2 = half ? az.zw : az.xy, in2 = half ? ain.zw : ain.xy;
    const v2d hn2 = half ? ahn.zw : ahn.xy, ho2 = half ? hold.zw : hold.xy;
    v2d rg, zg;
    sigmoid_pair2(r2, z2, &rg, &zg);
    const v2d ng = tanh2(fma2(rg, hn2, in2));
    const v2d hv = fma2(zg, ho2 - ng, ng);  // (h - n) z + n, the form aten's gru_cell evaluates (= (1-z) n + z h)
    if (half) {
      hnew.zw = hv;
    } else {
      hnew.xy = hv;
    }
  }
  return hnew;
}

// One wavefront encodes the 16 windows its lanes' columns name.  Mode 0: window row `wc` of the explicit (N, B, nin)
// tensor; mode 1: window (kk, tt) of the MPPI history.  H0 / H1: this wave's hidden-state images in LDS (KS*64 doubles
// each, every entry written and read by the same lane: no barrier).  Returns linear_out row q (valid for q < 2).
template <int G>
__device__ __forceinline__ double gru_encode_tile(const GruArgs& a, int lane, int64_t wc, int64_t kk, int tt,
                                                  double* __restrict__ H0, double* __restrict__ H1) {
  constexpr int GT = G / 16;   // tiles per gate = chunks
  constexpr int KS = G / 4;    // k-steps over the hidden dimension
  const int q = lane >> 4;
  // per-lane input normalisation constants (lane q feeds input dim q; q == 3 feeds the bias column)
  double in_mean = 0.0, in_std = 1.0;
  if (q < a.nin) {
    in_mean = a.mean[q];
    in_std = a.std[q];
  }
  const int ab_off = (a.mode == 1) ? (int)(kk / a.Kep) * a.B : 0;  // this sample's episode block of action_buffer
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    H0[ks * 64 + lane] = 0.0;
    H1[ks * 64 + lane] = 0.0;
  }
  v4d hn[GT];
  // A non-finite window entry must leave as NaN latents, as nn.GRU's arithmetic gives it (the reference, w_nl.py:14-29): the
  // gate math here clamps with v_min / v_max, which swallow a NaN -- so the window is remembered and its outputs are replaced
  bool bad_input = false;

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
          raw = (double)(a.B - 1 - j_win);  // encode_obs_time model: the harness's constant time channel
      }
      xin = (raw - in_mean) / in_std;
      bad_input = bad_input || !(xin - xin == 0.0);
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
      if (s > 0) chunk_gemm<KS>(ar, az, ahn, a.Whh0p + (size_t)j * KS * 3 * 64, lane, H0);
      const v4d hold = {H0[(4 * j + 0) * 64 + lane], H0[(4 * j + 1) * 64 + lane], H0[(4 * j + 2) * 64 + lane],
                        H0[(4 * j + 3) * 64 + lane]};
      hn[j] = gru_gates(ar, az, ain, ahn, hold);
    }
#pragma unroll
    for (int j = 0; j < GT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) H0[(4 * j + r) * 64 + lane] = hn[j][r];
    // ---------------- layer 1
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      v4d ar = load_bias_tile(a.brz1, j, q);
      v4d az = load_bias_tile(a.brz1, GT + j, q);
      v4d ain = load_bias_tile(a.bin1, j, q);
      v4d ahn = load_bias_tile(a.bhn1, j, q);
      chunk_gemm<KS>(ar, az, ain, a.Wih1p + (size_t)j * KS * 3 * 64, lane, H0);
      if (s > 0) chunk_gemm<KS>(ar, az, ahn, a.Whh1p + (size_t)j * KS * 3 * 64, lane, H1);
      const v4d hold = {H1[(4 * j + 0) * 64 + lane], H1[(4 * j + 1) * 64 + lane], H1[(4 * j + 2) * 64 + lane],
                        H1[(4 * j + 3) * 64 + lane]};
      hn[j] = gru_gates(ar, az, ain, ahn, hold);
    }
#pragma unroll
    for (int j = 0; j < GT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) H1[(4 * j + r) * 64 + lane] = hn[j][r];
  }
  // ---------------- linear_out (2 x g): rows 0,1 of one output tile
  v4d o[1];
  o[0] = splat(0.0);
  gemm_acc<1, KS>(o, a.Wop, lane, [&](int ks) { return H1[ks * 64 + lane]; });
  return nan_if_bad_window(bad_input, o[0][0] + a.bo[q < 2 ? q : 0]);
}

// ---------------------------------------------------------------------------------------------------------------
// COOPERATIVE tile: ONE workgroup of four wavefronts encodes ONE 16-window tile, wave w owning the gate chunks w, w + 4,
// ... (16 hidden units each) of both layers -- G = 64: one chunk per wave; G = 128: two; G = 32: waves 0 and 1 own one,
// the other two only keep the barriers.  Same arithmetic per entry as gru_encode_tile (same chunk GEMMs in the same k
// order, same gate math): bit-identical results, a fraction of the latency per tile.  The hidden-state images are shared
// by the workgroup and double-buffered: Hc[(2 l + b) * KS*64 + ks*64 + lane], layer l, buffer b; one barrier per GRU step
// between the layers and one before the head.  All four waves call this with the same (wc, kk, tt); wave 0 returns
// linear_out row q (valid for q < 2), the others return 0.
// Where a window's raw (un-normalised) inputs come from is a policy of the cooperative tile:
//   XDirect  the explicit window tensor (mode 0) / the MPPI history over the perturbed-action tensor an EARLIER launch
//            wrote (mode 1) -- gru_encode_coop_kernel, and the fused planner body behind a perturb kernel
//   (kernels_fused.hip) XInline: the fused planner body samples and bounds the actions itself
// Interface: prepare() is called by all four waves before the tile's first barrier; raw() after it, by lanes q < nin.
struct XDirect {
  __device__ __forceinline__ void prepare(const GruArgs&, int, int, int64_t, int, bool) {}
  __device__ __forceinline__ double raw(const GruArgs& a, int64_t wc, int64_t kk, int tt, int j_win, int q, int c,
                                        int ab_off) const {
    if (a.mode == 0) return a.window[(wc * a.B + j_win) * a.nin + q];
    const int i = tt + j_win;
    if (q < a.nact)
      return (i < a.B - 1) ? a.abuf[(ab_off + 1 + i) * a.nact + q]
                           : a.u_scale * a.perturbed[(kk * a.T + (i - (a.B - 1))) * a.nact + q];
    return (double)(a.B - 1 - j_win);  // encode_obs_time model: the harness's constant time channel
  }
};

template <int G, class XS = XDirect>
__device__ __forceinline__ double gru_encode_tile_coop(const GruArgs& a, int lane, int wv, int64_t wc, int64_t kk, int tt,
                                                       double* __restrict__ Hc, XS xs = XS(), bool valid = true) {
  constexpr int GT = G / 16;
  constexpr int KS = G / 4;
  constexpr int CPW = (GT + 3) / 4;  // chunks per wave
  constexpr bool kAllWaves = GT % 4 == 0;  // every wave owns CPW chunks: no guard (a guard the compiler cannot fold costs
                                           // the G = 64 kernel 95 spilled VGPRs)
  constexpr int IMG = KS * 64;       // one image; layer-0 buffers at Hc + {0, IMG}, layer-1 buffers at Hc + {2, 3} IMG
  const int q = lane >> 4;
  double in_mean = 0.0, in_std = 1.0;
  if (q < a.nin) {
    in_mean = a.mean[q];
    in_std = a.std[q];
  }
  const int ab_off = (a.mode == 1) ? (int)(kk / a.Kep) * a.B : 0;
  xs.prepare(a, lane, wv, kk, tt, valid);
  // zero this wave's rows of the "current" images (h_0 = 0); the other buffers are written before they are read
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int j = wv + 4 * i;
    if (kAllWaves || j < GT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Hc[(4 * j + r) * 64 + lane] = 0.0;
        Hc[2 * IMG + (4 * j + r) * 64 + lane] = 0.0;
      }
    }
  }
  __syncthreads();
  bool bad_input = false;  // (see gru_encode_tile)
  for (int s = 0; s < a.B; ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
    double* H0c = Hc + cur * IMG;
    double* H0n = Hc + nxt * IMG;
    double* H1c = Hc + (2 + cur) * IMG;
    double* H1n = Hc + (2 + nxt) * IMG;
    const int j_win = a.B - 1 - s;
    double xin = 0.0;
    if (q < a.nin) {
      const double raw = xs.raw(a, wc, kk, tt, j_win, q, lane & 15, ab_off);
      xin = (raw - in_mean) / in_std;
      bad_input = bad_input || !(xin - xin == 0.0);
    } else if (q == 3) {
      xin = 1.0;
    }
    // ---------------- layer 0, this wave's chunks (the new state goes to the OTHER buffer: no chunk waits for another)
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int j = wv + 4 * i;
      if (kAllWaves || j < GT) {
        gptr wp = opaque(a.Wih0p + (size_t)j * 3 * 64);
        v4d ar = mfma(wp[lane], xin, splat(0.0));
        v4d az = mfma(wp[64 + lane], xin, splat(0.0));
        v4d ain = mfma(wp[128 + lane], xin, splat(0.0));
        v4d ahn = load_bias_tile(a.bhn0, j, q);
        if (s > 0) chunk_gemm<KS>(ar, az, ahn, a.Whh0p + (size_t)j * KS * 3 * 64, lane, H0c);
        const v4d hold = {H0c[(4 * j + 0) * 64 + lane], H0c[(4 * j + 1) * 64 + lane], H0c[(4 * j + 2) * 64 + lane],
                          H0c[(4 * j + 3) * 64 + lane]};
        const v4d hn = gru_gates(ar, az, ain, ahn, hold);
#pragma unroll
        for (int r = 0; r < 4; ++r) H0n[(4 * j + r) * 64 + lane] = hn[r];
      }
    }
    __syncthreads();  // layer 0's new state is complete; everybody has read the old one
    // ---------------- layer 1
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int j = wv + 4 * i;
      if (kAllWaves || j < GT) {
        v4d ar = load_bias_tile(a.brz1, j, q);
        v4d az = load_bias_tile(a.brz1, GT + j, q);
        v4d ain = load_bias_tile(a.bin1, j, q);
        v4d ahn = load_bias_tile(a.bhn1, j, q);
        chunk_gemm<KS>(ar, az, ain, a.Wih1p + (size_t)j * KS * 3 * 64, lane, H0n);
        if (s > 0) chunk_gemm<KS>(ar, az, ahn, a.Whh1p + (size_t)j * KS * 3 * 64, lane, H1c);
        const v4d hold = {H1c[(4 * j + 0) * 64 + lane], H1c[(4 * j + 1) * 64 + lane], H1c[(4 * j + 2) * 64 + lane],
                          H1c[(4 * j + 3) * 64 + lane]};
        const v4d hn = gru_gates(ar, az, ain, ahn, hold);
#pragma unroll
        for (int r = 0; r < 4; ++r) H1n[(4 * j + r) * 64 + lane] = hn[r];
      }
    }
    // (no barrier here: the next step's layer 0 touches only the layer-0 images, and its barrier orders this step's
    // layer-1 writes before the next step's layer-1 reads)
  }
  __syncthreads();  // the last layer-1 state is complete
  double out = 0.0;
  if (wv == 0) {
    const double* Hl = Hc + (2 + (a.B & 1)) * IMG;
    v4d o[1];
    o[0] = splat(0.0);
    gemm_acc<1, KS>(o, a.Wop, lane, [&](int ks) { return Hl[ks * 64 + lane]; });
    out = o[0][0] + a.bo[q < 2 ? q : 0];
  }
  return nan_if_bad_window(bad_input, out);  // (every wave sees the same window inputs; wave 0's value is the one stored)
}

}  // namespace nlc
