// ReverseGRUEncoder (w_nl.py:14-29) on FP64 matrix cores.
//
// One wavefront encodes 16 action windows.  Hidden state and gate pre-activations live in MFMA
// accumulator layout (feature on rows/registers, window on columns/lanes): accumulator register r of tile j IS
// the B fragment of k-step 4j+r of the next GEMM, so the recurrence h -> gates -> h' needs no lane movement.
// The two layers' hidden states are parked in LDS as per-lane images of exactly those B fragments (each entry
// written and read by the same lane: no barrier), which keeps the kernel at ~200 VGPRs = 2 waves/SIMD with room
// for two-wide gate math.  Weights are the MFMA A operand, streamed fragment-packed from L2; the 16 windows are
// the B/D columns.  The encoder input does not depend on the state
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
__device__ __forceinline__ v2d rcp_refined2(v2d d) {
  // v_rcp_f64 (~2^-24) + one cubic step r (1 + e + e^2), e = 1 - d r: 3 FMAs, <= 1 ulp (tools/rcp_probe.hip)
  const v2d r = {__builtin_amdgcn_rcp(d.x), __builtin_amdgcn_rcp(d.y)};
  const v2d e = fma2(-d, r, splat2(1.0));
  return fma2(r, fma2(e, e, e), r);
}
__device__ __forceinline__ v2d expm1_poly2(v2d r) {
  v2d q = splat2(0x1.af38a9b0ec855p-26);
  q = fma2(q, r, splat2(0x1.289185613a3d6p-22));
  q = fma2(q, r, splat2(0x1.71de0dae63bb3p-19));
  q = fma2(q, r, splat2(0x1.a019b90d2ae7ap-16));
  q = fma2(q, r, splat2(0x1.a01a01a7c41d5p-13));
  q = fma2(q, r, splat2(0x1.6c16c1788bd90p-10));
  q = fma2(q, r, splat2(0x1.11111111109b3p-7));
  q = fma2(q, r, splat2(0x1.5555555553d63p-5));
  q = fma2(q, r, splat2(0x1.5555555555556p-3));
  q = fma2(q, r, splat2(0x1.0000000000001p-1));
  return fma2(q * r, r, r);
}
__device__ __forceinline__ v2d exp_reduce2(v2d y, v2i* n) {
  // round-to-nearest by the 1.5 * 2^52 shift; the integer is the low word of the shifted sum (no v_rndne / v_cvt)
  const v2d sh = fma2(y, splat2(1.44269504088896338700e+00), splat2(6755399441055744.0));
  const v2d fn = sh - splat2(6755399441055744.0);
  v2d r = fma2(-fn, splat2(6.93147180369123816490e-01), y);
  r = fma2(-fn, splat2(1.90821492927058770002e-10), r);
  *n = v2i{__double2loint(sh.x), __double2loint(sh.y)};
  return r;
}
// 1 + e^{-x} for either sign (e^{-x} >= 0, nothing cancels).  The argument is clamped to +-350 so that the product
// of two such terms stays finite (below -350 the true sigmoid is < 1e-152 and this returns ~1e-152).
__device__ __forceinline__ v2d one_plus_exp_neg2(v2d x) {
  const v2d y = __builtin_elementwise_min(__builtin_elementwise_max(-x, splat2(-350.0)), splat2(350.0));
  v2i n;
  const v2d r = exp_reduce2(y, &n);
  const v2d p = splat2(1.0) + expm1_poly2(r);
  const v2d e = {ldexp(p.x, n.x), ldexp(p.y, n.y)};
  return splat2(1.0) + e;
}
// the reset and update gates share ONE reciprocal: sigmoid(a) = B / (A B), sigmoid(b) = A / (A B) with
// A = 1 + e^{-a}, B = 1 + e^{-b}  (three multiplies instead of a second v_rcp_f64 + refinement)
__device__ __forceinline__ void sigmoid_pair2(v2d a, v2d b, v2d* sa, v2d* sb) {
  const v2d A = one_plus_exp_neg2(a), B = one_plus_exp_neg2(b);
  const v2d R = rcp_refined2(A * B);
  *sa = B * R;
  *sb = A * R;
}
__device__ __forceinline__ v2d tanh2(v2d x) {
  const v2d y = __builtin_elementwise_max(-2.0 * __builtin_elementwise_abs(x), splat2(-745.0));
  v2i n;
  const v2d r = exp_reduce2(y, &n);
  const v2d p = expm1_poly2(r);
  const v2d two_n = {ldexp(1.0, n.x), ldexp(1.0, n.y)};
  const v2d em = fma2(two_n, p, two_n - splat2(1.0));
  const v2d t = -em * rcp_refined2(splat2(2.0) + em);
  return __builtin_elementwise_copysign(t, x);
}

// DBG = 1: timing experiment only (env NLC_GRU_DBG=1): gates replaced by a few FMAs -> MFMA + load time
template <int DBG>
__device__ __forceinline__ v4d gru_gates(const v4d& ar, const v4d& az, const v4d& ain, const v4d& ahn, const v4d& hold) {
  v4d hnew;
  if (DBG == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) hnew[r] = 0.25 * ar[r] + 0.125 * az[r] + 0.01 * (ain[r] + ahn[r]) + 0.5 * hold[r];
    return hnew;
  }
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
  const int ab_off = (a.mode == 1) ? (int)(kk / a.Kep) * a.B : 0;  // this sample's episode block of action_buffer

  // Hidden states live in LDS as per-lane B-fragment images (H[ks*64 + lane], ks = 4*tile + reg): every entry is
  // written and read by the same lane, so there is no cross-lane hazard and no barrier; registers only hold the
  // chunk accumulators and the new state being assembled.
  __shared__ double Hs[4][2][KS * 64];
  double* H0 = Hs[wave][0];
  double* H1 = Hs[wave][1];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    H0[ks * 64 + lane] = 0.0;
    H1[ks * 64 + lane] = 0.0;
  }
  v4d hn[GT];

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
      hn[j] = gru_gates<DBG>(ar, az, ain, ahn, hold);
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
      hn[j] = gru_gates<DBG>(ar, az, ain, ahn, hold);
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

// ------------------------------------------------------------------ Delta-t RNN baseline
// DeltaTRNN (train_utils.py:589-631): ONE GRU layer of hidden size G read in FORWARD order over the action window,
// then linear_out over [h_last | obs | ts].  The window does not depend on the state, so -- exactly as for the NL
// encoder above -- the planner hoists the GRU out of the horizon loop and this kernel also applies the hidden part
// of linear_out: q = W_out[:, :G] h_last (d values per window), leaving a d x d matvec per horizon step to the
// rollout kernel.  Same dataflow as gru_encode_kernel (weights = MFMA A operand, 16 windows = B/D columns, hidden
// state parked in LDS as per-lane B-fragment images).
// Roofline: FP64 MFMA.  Per 16 windows: B*3*GT (input) + (B-1)*3*GT*KS (hidden) + KS (head) MFMAs of 2048 flop;
// G = 160, B = 4: 3760 MFMAs = 481 kflop per window.  HBM: nu*8 B in, d*8 B out per window.
// (G = 64 fits 2 waves/SIMD; at G >= 128 the new state alone is 64-80 VGPRs and the kernel runs one wave per SIMD --
// forcing 256 registers spills 116-210 of them)
template <int G, int WPB>
__global__ __launch_bounds__(64 * WPB, G <= 64 ? 2 : 1) void rnn_encode_kernel(const RnnArgs a) {
  constexpr int GT = G / 16;
  constexpr int KS = G / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, c = lane & 15;
  const int64_t w = ((int64_t)blockIdx.x * WPB + wave) * 16 + c;
  const bool valid = w < a.N;
  const int64_t wc = valid ? w : a.N - 1;
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
  const int ab_off = (a.mode == 1) ? (int)(kk / a.Kep) * a.B : 0;
  __shared__ double Hs[WPB][KS * 64];
  double* H0 = Hs[wave];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) H0[ks * 64 + lane] = 0.0;
  v4d hn[GT];
  for (int s = 0; s < a.B; ++s) {
    // forward time: GRU step s consumes window element s
    double xin = 0.0;
    if (q < a.nin) {
      double raw;
      if (a.mode == 0) {
        raw = a.window[(wc * a.B + s) * a.nin + q];
      } else {
        const int i = tt + s;  // hist = [action_buffer[1:] ; u_scale * perturbed]  (mppi_delay.py:254-260)
        raw = (i < a.B - 1) ? a.abuf[(ab_off + 1 + i) * a.nin + q]
                            : a.u_scale * a.perturbed[(kk * a.T + (i - (a.B - 1))) * a.nin + q];
      }
      xin = (raw - in_mean) / in_std;
    } else if (q == 3) {
      xin = 1.0;  // bias column of the packed W_ih
    }
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      gptr wp = opaque(a.Wihp + (size_t)j * 3 * 64);
      v4d ar = mfma(wp[lane], xin, splat(0.0));
      v4d az = mfma(wp[64 + lane], xin, splat(0.0));
      v4d ain = mfma(wp[128 + lane], xin, splat(0.0));
      v4d ahn = load_bias_tile(a.bhn, j, q);
      if (s > 0) chunk_gemm<KS>(ar, az, ahn, a.Whhp + (size_t)j * KS * 3 * 64, lane, H0);
      const v4d hold = {H0[(4 * j + 0) * 64 + lane], H0[(4 * j + 1) * 64 + lane], H0[(4 * j + 2) * 64 + lane],
                        H0[(4 * j + 3) * 64 + lane]};
      hn[j] = gru_gates<0>(ar, az, ain, ahn, hold);
    }
#pragma unroll
    for (int j = 0; j < GT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) H0[(4 * j + r) * 64 + lane] = hn[j][r];
  }
  // hidden part of linear_out (d <= 8 rows of one output tile): register r of lane group q holds row 4 r + q
  v4d o[1];
  o[0] = splat(0.0);
  gemm_acc<1, KS>(o, a.Wop, lane, [&](int ks) { return H0[ks * 64 + lane]; });
  if (valid) {
    const int64_t row = (a.mode == 1) ? (int64_t)tt * a.K + kk : w;
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (4 * r + q < a.d) a.out[row * a.d + 4 * r + q] = o[0][r];
  }
}

hipError_t launch_rnn_encode(const RnnArgs& a, int hidden, hipStream_t s) {
  if (a.N <= 0) return hipSuccess;
  // two waves per workgroup: the G = 160 hidden-state images are 20 KB per wave
  const unsigned grid = (unsigned)((a.N + 31) / 32);
  if (hidden == 160) {
    hipLaunchKernelGGL((rnn_encode_kernel<160, 2>), dim3(grid), dim3(128), 0, s, a);
  } else if (hidden == 128) {
    hipLaunchKernelGGL((rnn_encode_kernel<128, 2>), dim3(grid), dim3(128), 0, s, a);
  } else if (hidden == 64) {
    hipLaunchKernelGGL((rnn_encode_kernel<64, 2>), dim3(grid), dim3(128), 0, s, a);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace nlc
