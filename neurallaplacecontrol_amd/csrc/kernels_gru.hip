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
#include "nlc_device.h"
#include "nlc_gru_tile.h"
#include "nlc_kernels.h"

namespace nlc {

// (G = 128, hidden_units = 256: the two layers' images are 128 KB of LDS for the four waves -- one workgroup per CU, one
// wave per SIMD, as for the baselines' rnn_encode_kernel<128>)
template <int G>
__global__ __launch_bounds__(256, G <= 64 ? 2 : 1) void gru_encode_kernel(const GruArgs a) {
  constexpr int KS = G / 4;    // k-steps over the hidden dimension
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
  // Hidden states live in LDS as per-lane B-fragment images (H[ks*64 + lane], ks = 4*tile + reg): every entry is
  // written and read by the same lane, so there is no cross-lane hazard and no barrier; registers only hold the
  // chunk accumulators and the new state being assembled.
  __shared__ double Hs[4][2][KS * 64];
  const double o = gru_encode_tile<G>(a, lane, wc, kk, tt, Hs[wave][0], Hs[wave][1]);
  if (valid && q < 2) {
    const int64_t wo = (a.mode == 1) ? kk * a.T + tt : w;  // (k, t) -> row of the (K, T, 2) latent tensor
    a.out[wo * 2 + q] = o;
  }
}

// Cooperative variant (one 16-window tile per workgroup, one gate chunk per wave: nlc_gru_tile.h): same results, a
// quarter of the per-tile latency -- for launches with too few tiles to fill the chip with wave-sized ones.
template <int G>
__global__ __launch_bounds__(256, G <= 64 ? 4 : 2) void gru_encode_coop_kernel(const GruArgs a) {
  constexpr int KS = G / 4;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, c = lane & 15;
  __shared__ double Hc[4 * KS * 64];
  for (int64_t tile = blockIdx.x; tile * 16 < a.N; tile += gridDim.x) {
    const int64_t w = tile * 16 + c;
    const bool valid = w < a.N;
    const int64_t wc = valid ? w : a.N - 1;
    int64_t kk = 0;
    int tt = 0;
    if (a.mode == 1) {
      kk = wc / a.Tc;
      tt = a.t0 + (int)(wc - kk * a.Tc);
    }
    const double o = gru_encode_tile_coop<G>(a, lane, wv, wc, kk, tt, Hc);
    if (wv == 0 && valid && q < 2) {
      const int64_t wo = (a.mode == 1) ? kk * a.T + tt : w;
      a.out[wo * 2 + q] = o;
    }
    __syncthreads();  // the images are re-zeroed by the next tile
  }
}

// lds_pad_bytes (cooperative form only): unused dynamic LDS that lowers the kernel's occupancy, for launches that are
// meant to SHARE the CUs with another stream's kernels (the staged de Hoog planner encodes later horizon chunks beside its
// step chain: two encoder workgroups per CU instead of four leave registers and issue slots for the chain's kernels)
hipError_t launch_gru_encode(const GruArgs& a, int g, hipStream_t s, bool coop, unsigned lds_pad_bytes) {
  if (a.N <= 0) return hipSuccess;
  // `gru_gemm = 1`: the int8-sliced encoder takes the launches the wave-sized form would get (auto: more than 50 000 windows; below
  // that the cooperative FP64 form's latency wins -- 0.115 vs 0.133 ms at 20 480 windows, level at 40 960, 0.395 vs 0.329 at 81 920;
  // `gru_coop = 0` sends every launch here)
  if (g == 64 && a.use_i8 && !coop) return launch_gru_encode_i8(a, s);
  if (coop) {
    const int64_t tiles = (a.N + 15) / 16;
    const dim3 cgrid((unsigned)(tiles < 65536 ? tiles : 65536));
    if (g == 64) {
      hipLaunchKernelGGL(gru_encode_coop_kernel<64>, cgrid, dim3(256), lds_pad_bytes, s, a);
    } else if (g == 32) {
      hipLaunchKernelGGL(gru_encode_coop_kernel<32>, cgrid, dim3(256), lds_pad_bytes, s, a);
    } else if (g == 128) {
      hipLaunchKernelGGL(gru_encode_coop_kernel<128>, cgrid, dim3(256), lds_pad_bytes, s, a);
    } else {
      return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  const unsigned grid = (unsigned)((a.N + 63) / 64);
  if (g == 64) {
    hipLaunchKernelGGL((gru_encode_kernel<64>), dim3(grid), dim3(256), 0, s, a);
  } else if (g == 32) {
    hipLaunchKernelGGL((gru_encode_kernel<32>), dim3(grid), dim3(256), 0, s, a);
  } else if (g == 128) {
    hipLaunchKernelGGL((gru_encode_kernel<128>), dim3(grid), dim3(256), 0, s, a);
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
      hn[j] = gru_gates(ar, az, ain, ahn, hold);
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
