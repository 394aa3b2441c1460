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
#include <atomic>

#include "nlc_device.h"
#include "nlc_gru_tile.h"
#include "nlc_i8gemm.h"
#include "nlc_kernels.h"

namespace nlc {

// ---- the weight stream.  A GRU step consumes 36 gate tiles in a fixed order (kI8Seq* below); every workgroup reads the same
// 36 x 7.25 KB, so its four wavefronts fetch each block ONCE, two tiles ahead of its use, and pass it through LDS: the loads'
// latency (L2: the stream is 261 KB) is off the waves' critical path and the L1 -> register traffic is a quarter of four private
// streams.  (First version: every wave loaded its own fragments right before their MFMAs -- 44 % of all wave cycles parked in
// s_waitcnt, 3.36 ms against the FP64 kernel's 2.99; profiles/r5_i8_gemm.md.)
// Block layout (nlc_pack.h: pack_gru_i8_stream): seven digit fragments [digit][lane][16 B], then the tile's 16 recombination
// factors and its 16 biases (feature order of the tile, f = 4 r + q).
constexpr int kI8DigitBytes = i8::kDigits * 64 * 16;   // 7168
constexpr int kI8Block = kI8DigitBytes + 2 * 16 * 8;   // 7424 = 232 x 32
constexpr int kI8StageThreads = kI8Block / 32;
struct WeightStream {
  const signed char* base;  // 36 blocks, consumption order of a step s >= 1
  char* lds;                // 2 x kI8Block
  int tid;
  i8::v4i st0, st1;         // this thread's 32 bytes of the block after next
  __device__ __forceinline__ void fetch(int t) {
    typedef const __attribute__((address_space(1))) i8::v4i* g4;
    const int off = (tid < kI8StageThreads ? tid : kI8StageThreads - 1) * 32;  // (no divergent load: the last threads re-read)
    g4 p = (g4)(base + (size_t)t * kI8Block + off);
    st0 = p[0];
    st1 = p[1];
  }
  __device__ __forceinline__ void put(int buf) {
    if (tid < kI8StageThreads) {
      i8::v4i* d = (i8::v4i*)(lds + buf * kI8Block + tid * 32);
      d[0] = st0;
      d[1] = st1;
    }
  }
};
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// position p of step 0 (h = 0: only layer 1's input-side tiles, 12 of them) -> block; step s >= 1 consumes blocks 0 .. 35 in order
// (layer 1: the hidden-side tile of a gate BEFORE its input-side tile, see gru_encode_tile_i8)
__device__ __forceinline__ constexpr int i8_seq0(int p) { return 12 + 6 * (p / 3) + 2 * (p % 3) + 1; }
__device__ __forceinline__ constexpr int i8_next2_step0(int p) { return p + 2 < 12 ? i8_seq0(p + 2) : p + 2 - 12; }
__device__ __forceinline__ constexpr int i8_next2(int p) { return (p + 2) % 36; }

// one tile of the stream: hand the staged block on, fetch the block after next, run `body` on the current block, barrier
template <class F>
__device__ __forceinline__ void i8_op(WeightStream& ws, int parity, int next2, F body) {
#if NLC_I8_DBG == 3  // tools only (timing): no staging, no barrier -- every tile reads the first block (wrong results)
  body((const char*)ws.lds);
  return;
#endif
  ws.put(parity ^ 1);
  ws.fetch(next2);
  body((const char*)(ws.lds + parity * kI8Block));
  lds_barrier();
}
__device__ __forceinline__ void i8_read_digits(i8::v4i (&a)[i8::kDigits], const char* tb, int lane) {
#pragma unroll
  for (int i = 0; i < i8::kDigits; ++i) a[i] = *(const i8::v4i*)(tb + (i * 64 + lane) * 16);
}
__device__ __forceinline__ v4d i8_read_tile16(const char* tb, int which, int q) {  // which: 0 factors, 1 biases; register r <-> f = 4 r + q
  const double* p = (const double*)(tb + kI8DigitBytes + which * 128) + q;
  return v4d{p[0], p[4], p[8], p[12]};
}
// issue order of a tile whose MFMAs run beside independent VALU work of the same wave (the previous chunk's gate math): one
// MFMA, then `per` VALU instructions in its shadow -- the i8 MFMA occupies the matrix pipe for 16 clocks and the vector issue
// for 4 (tools/ubench_i8emu.hip: 8 MFMAs + 16 FP64 FMAs of one wave take 156 clocks against 136 + 100)
template <int PER>
__device__ __forceinline__ void i8_mfma_valu_order() {
  constexpr int n = i8::pair_order().n;
#pragma unroll
  for (int m = 0; m < n; ++m) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);
  }
}

// Software pipeline of a GRU step: the gate math of chunk j (260 FP64 VALU instructions per lane, the update of its state image)
// runs in the shadow of the FIRST tile of chunk j + 1 -- that tile's 34 MFMAs do not depend on it: within a layer every GEMM reads
// the OLD state's digits; across layers, layer 1 starts with a hidden-side tile (old h1) while layer 0's last chunk finishes and
// the new h0 is cut into digits, and the next step's layer 0 starts (old h0) while layer 1's last chunk finishes.
template <int G>
__device__ __forceinline__ double gru_encode_tile_i8(const GruArgs& a, WeightStream& ws, int lane, int64_t wc, int64_t kk, int tt,
                                                     double* __restrict__ H0, double* __restrict__ H1) {
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
  // digits of the two layers' states, dig[i][c] = digit i of the lane's entries 4 c .. 4 c + 3
  i8::v4i S0[i8::kDigits], S1[i8::kDigits];
  // Fixed point has no NaN / infinity (i8::fixq yields finite garbage digits): a window with a non-finite entry is remembered
  // here and its latents leave as NaN -- as in the FP64 kernels (nlc_gru_tile.h) and as nn.GRU gives it
  bool bad_input = false;
  auto window_input = [&](int s) {
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
      bad_input = bad_input || !(xin - xin == 0.0);
    } else if (q == 3) {
      xin = 1.0;
    }
    return xin;
  };
  // the pre-activations of the chunk whose gate math is pending: reset, update, n input side, n hidden side
  v4d pre[4];
  auto finish = [&](double* __restrict__ H, int j) {  // gates of the pending chunk j, update of its rows of the state image
    const v4d hold = {H[(4 * j + 0) * 64 + lane], H[(4 * j + 1) * 64 + lane], H[(4 * j + 2) * 64 + lane], H[(4 * j + 3) * 64 + lane]};
    // (tried: both halves of the gate math side by side, four chains in lockstep behind empty asms -- no change, 2.49 ms: the
    // partner wave covers the FP64 latency)
#if NLC_I8_DBG == 4  // tools only (timing): no gate math
    const v4d hn = pre[0] + pre[1] + pre[2] + pre[3] + hold;
#else
    const v4d hn = gru_gates(pre[0], pre[1], pre[2], pre[3], hold);
#endif
#pragma unroll
    for (int r = 0; r < 4; ++r) H[(4 * j + r) * 64 + lane] = hn[r];
  };
  auto digits_of = [&](i8::v4i (&S)[i8::kDigits], const double* __restrict__ H) {  // the whole state, from its image
#if NLC_I8_DBG == 5  // tools only (timing): no digit cut
    return;
#endif
#pragma unroll
    for (int j = 0; j < GT; ++j)
      i8::slice_chunk(S, j, v4d{H[(4 * j + 0) * 64 + lane], H[(4 * j + 1) * 64 + lane], H[(4 * j + 2) * 64 + lane], H[(4 * j + 3) * 64 + lane]});
  };
  auto zero = [](i8::v4i (&acc)[i8::kLevels]) {
#pragma unroll
    for (int l = 0; l < i8::kLevels; ++l) acc[l] = i8::v4i{0, 0, 0, 0};
  };
  // the stream's first two blocks
  ws.fetch(i8_seq0(0));
  ws.put(0);
  ws.fetch(i8_seq0(1));
  lds_barrier();

  // ================= step 0: both states are 0 -- layer 0 has no hidden-state GEMM, layer 1 only its input side (12 tiles)
  {
    const double xin = window_input(0);
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      gptr wp = opaque(a.Wih0p + (size_t)j * 3 * 64);
      const v4d ar = mfma(wp[lane], xin, splat(0.0));
      const v4d az = mfma(wp[64 + lane], xin, splat(0.0));
      const v4d ain = mfma(wp[128 + lane], xin, splat(0.0));
      const v4d ahn = load_bias_tile(a.bhn0, j, q);
      const v4d hn = gru_gates(ar, az, ain, ahn, splat(0.0));
#pragma unroll
      for (int r = 0; r < 4; ++r) H0[(4 * j + r) * 64 + lane] = hn[r];
      i8::slice_chunk(S0, j, hn);
    }
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      v4d nxt[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const int p = 3 * j + g;
        i8_op(ws, p & 1, i8_next2_step0(p), [&](const char* tb) {
          i8::v4i w[i8::kDigits], acc[i8::kLevels];
          i8_read_digits(w, tb, lane);
          zero(acc);
          i8::tile_mfma(acc, w, S0);
          if (g == 0 && j > 0) {
            finish(H1, j - 1);
            i8_mfma_valu_order<2>();
          }
          if (g < 2)  // (reset / update blocks carry the factors of the level-by-level sum)
            nxt[g] = i8::recombine<false>(acc, i8_read_tile16(tb, 0, q), i8_read_tile16(tb, 1, q));
          else
            nxt[g] = i8::recombine<true>(acc, i8_read_tile16(tb, 0, q), i8_read_tile16(tb, 1, q));
        });
      }
      pre[0] = nxt[0];
      pre[1] = nxt[1];
      pre[2] = nxt[2];
      pre[3] = load_bias_tile(a.bhn1, j, q);  // h1 = 0: the hidden side of n is its bias
    }
    // (layer 1's chunk 3 is pending)
  }
  // ================= steps 1 .. B - 1: 36 tiles each
  for (int s = 1; s < a.B; ++s) {
    const double xin = window_input(s);
    // ---------------- layer 0: input side one FP64 k-step (as gru_encode_tile), hidden side on the i8 pipe
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      v4d in[3], nxt[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const int p = 3 * j + g;
        i8_op(ws, p & 1, i8_next2(p), [&](const char* tb) {
          i8::v4i w[i8::kDigits], acc[i8::kLevels];
          i8_read_digits(w, tb, lane);
          zero(acc);
          i8::tile_mfma(acc, w, S0);
          if (g == 0) {
            // in the MFMAs' shadow: the pending chunk (the previous step's last chunk of layer 1, or this layer's chunk j - 1)
            if (j == 0) {
              finish(H1, GT - 1);
              digits_of(S1, H1);
            } else {
              finish(H0, j - 1);
            }
            i8_mfma_valu_order<2>();
            gptr wp = opaque(a.Wih0p + (size_t)j * 3 * 64);
            in[0] = mfma(wp[lane], xin, splat(0.0));
            in[1] = mfma(wp[64 + lane], xin, splat(0.0));
            in[2] = mfma(wp[128 + lane], xin, splat(0.0));
          }
          nxt[g] = i8::recombine<true>(acc, i8_read_tile16(tb, 0, q), g < 2 ? in[g] : i8_read_tile16(tb, 1, q));
        });
      }
      pre[0] = nxt[0];
      pre[1] = nxt[1];
      pre[2] = in[2];
      pre[3] = nxt[2];
    }
    // ---------------- layer 1: per chunk and gate the hidden-side tile (old h1), then the input-side tile (new h0)
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      v4d nxt[4];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const int p = 12 + 6 * j + 2 * g;
        i8::v4i acc[i8::kLevels];
        zero(acc);
        i8_op(ws, p & 1, i8_next2(p), [&](const char* tb) {
          i8::v4i w[i8::kDigits];
          i8_read_digits(w, tb, lane);
          i8::tile_mfma(acc, w, S1);
          if (g == 0) {
            if (j == 0) {
              finish(H0, GT - 1);
              digits_of(S0, H0);  // (complete before the next tile, the first to read the new h0)
            } else {
              finish(H1, j - 1);
            }
            i8_mfma_valu_order<2>();
          }
          if (g == 2) {
            nxt[3] = i8::recombine<true>(acc, i8_read_tile16(tb, 0, q), i8_read_tile16(tb, 1, q));
            zero(acc);
          }
        });
        i8_op(ws, (p + 1) & 1, i8_next2(p + 1), [&](const char* tb) {
          i8::v4i w[i8::kDigits];
          i8_read_digits(w, tb, lane);
          i8::tile_mfma(acc, w, S0);
          // reset / update: one accumulator for both GEMMs (the two tiles carry the same factors and the same bias)
          if (g < 2)
            nxt[g] = i8::recombine<false>(acc, i8_read_tile16(tb, 0, q), i8_read_tile16(tb, 1, q));
          else
            nxt[2] = i8::recombine<true>(acc, i8_read_tile16(tb, 0, q), i8_read_tile16(tb, 1, q));
        });
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) pre[g] = nxt[g];
    }
  }
  finish(H1, GT - 1);  // layer 1's last chunk
  // ---------------- linear_out (2 x g): rows 0,1 of one output tile, FP64
  v4d o[1];
  o[0] = splat(0.0);
  gemm_acc<1, KS>(o, a.Wop, lane, [&](int ks) { return H1[ks * 64 + lane]; });
  return nan_if_bad_window(bad_input, o[0][0] + a.bo[q < 2 ? q : 0]);
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
  // 64 KB of state images + 14.5 KB of weight staging: two workgroups per CU (160 KB)
  __shared__ double Hs[4][2][KS * 64];
  __shared__ __attribute__((aligned(16))) char Wst[2 * kI8Block];
#if NLC_I8_WAVES == 1
  __shared__ double pad_[4096];  // + 32 KB: a second workgroup does not fit the CU
  if (a.N < 0) pad_[threadIdx.x] = 0.0;
#endif
  WeightStream ws{a.i8_stream, Wst, (int)threadIdx.x, {}, {}};
  const double o = gru_encode_tile_i8<G>(a, ws, lane, wc, kk, tt, Hs[wave][0], Hs[wave][1]);
  if (valid && q < 2) {
    const int64_t wo = (a.mode == 1) ? kk * a.T + tt : w;
    a.out[wo * 2 + q] = o;
  }
}

// launches of the sliced kernel in this process (nlc_get_stat "gru_i8_launches": the GPU suite checks that a test in the fast mode
// really ran it -- the option alone does not say so, a cooperative or fused launch keeps its FP64 encoder)
static std::atomic<unsigned long long> g_i8_launches{0};
unsigned long long gru_i8_launch_count() { return g_i8_launches.load(); }

hipError_t launch_gru_encode_i8(const GruArgs& a, hipStream_t s) {
  if (a.N <= 0) return hipSuccess;
  g_i8_launches.fetch_add(1);
  const unsigned grid = (unsigned)((a.N + 63) / 64);
  hipLaunchKernelGGL(gru_encode_i8_kernel, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace nlc
