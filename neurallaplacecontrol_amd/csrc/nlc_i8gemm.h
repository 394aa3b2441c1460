// FP64 GEMMs with bounded operands on the INT8 matrix pipe (round 5, experimental: `gru_gemm = 1`).
//
// Why: v_mfma_f64_16x16x4_f64 holds the SIMD's vector issue for its 64 clocks, so the encoder's FP64 gate math ADDS to its MFMA
// time (0.72 + 0.24 of every SIMD cycle).  v_mfma_i32_16x16x64_i8 costs 9 clocks per SIMD at two waves, covers the whole
// K = 64 of a gate tile in one instruction and runs BESIDE the VALU (tools/ubench_i8emu.hip, profiles/r5_ubench_i8emu.txt).
//
// How: the GEMM operands here are bounded -- GRU states lie in [-1, 1], a weight row is scaled by a power of two to [-1, 1] --
// so both are written in FIXED POINT with kFrac = 54 fractional bits, X = rint(x 2^54), and cut into seven signed 8-bit digits
//   X = sum_i d_i 256^i,  d_0 .. d_5 in [-128, 127], d_6 the signed rest (|d_6| <= 65).
// W h = s_m 2^-108 sum_{i, j} 256^(i + j) (D_i . E_j): every digit product D_i . E_j (K = 64 terms of at most 2^14) is ONE i8
// MFMA, exact in int32; products of the same level L = i + j share an accumulator (at most 7 x 64 x 2^14 < 2^23); levels
// L < kLmin are dropped (kLmin = 5: at most 5 x 2^-56 s_m, a third of ONE operand's quantisation); the levels are recombined in
// FP64 (Horner, lowest first).  Operand error: 2^-55 absolute per entry (relative to the row scale for a weight) -- the FP64 MFMA
// path rounds its 64-term accumulation at 2^-53 of the running sum sixteen times.  tools/i8gemm_check.hip measures both against
// the exact product.
// Per 16 x 16 output tile and K = 64: 34 i8 MFMAs (~300 clocks of the matrix pipe) + ~60 VALU instructions per lane, against 16 FP64
// MFMAs (1 024 clocks of matrix AND vector issue).
#pragma once
#include "nlc_device.h"

namespace nlc {
namespace i8 {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kDigits = 7;
#ifndef NLC_I8_DBG
#define NLC_I8_DBG 0
#endif
#ifndef NLC_I8_LMIN
#define NLC_I8_LMIN 5
#endif
constexpr int kLmin = NLC_I8_LMIN;                        // digit pairs with i + j < kLmin are dropped
constexpr int kLevels = 2 * (kDigits - 1) - kLmin + 1;     // levels kLmin .. 12
constexpr int kTop = 2 * (kDigits - 1);                   // the highest level
constexpr int kFrac = 54;                                  // fractional bits of both operands' fixed point
// recombine() returns sum_L 256^(L - kTop) c_L; a row with scale s_m = 2^e carries the factor s_m 2^(-2 kFrac) 256^kTop = s_m 2^kRowExp2
constexpr int kRowExp2 = 8 * kTop - 2 * kFrac;
constexpr unsigned long long kBias = 0x0000808080808080ull;  // 128 in each of the six low digits

// X' = rint(x 2^kFrac) + kBias of |x| <= 1, as two dwords.  Four FP64 instructions + one 64-bit multiply-add: the two halves of X
// come out of the mantissa of a 1.5 2^52-shifted sum (round to nearest even, two's complement in the low dword); the bias rides
// in the shift constants.  NaN / infinity are not represented (the native path propagates them; this one yields garbage digits).
__device__ __forceinline__ void fixq(double x, unsigned& lo, unsigned& hi) {
#pragma clang fp contract(off)
  constexpr double kM = 6755399441055744.0;  // 1.5 2^52
  constexpr double kM1 = kM + 131586.0;      // + (kBias >> 30)
  constexpr double kM3 = kM + 8421504.0;     // + (kBias & (2^30 - 1))
  constexpr double kHi = (double)(1ll << (kFrac - 30));
  const double t1 = fma(x, kHi, kM1);            // x 2^24
  const double hif = t1 - kM1;                   // rint(x 2^24), exact
  const double r = fma(x, kHi, -hif);            // in [-1/2, 1/2], exact
  const double t3 = fma(r, 1073741824.0, kM3);   // r 2^30
  const long long X = (long long)__double2loint(t1) * (1ll << 30) + (long long)__double2loint(t3);
  lo = (unsigned)X ^ 0x80808080u;
  hi = (unsigned)((unsigned long long)X >> 32) ^ 0x00008080u;
}
// (tried: the four entries of a chunk stage by stage behind empty asms, as recombine() does -- 2.47 -> 2.76 ms: the slicing sits
// at the layer boundary, where the extra live values spill)

// 4 x 4 byte transpose: o[i] = (w[0].byte i, w[1].byte i, w[2].byte i, w[3].byte i), eight v_perm_b32
// (__builtin_amdgcn_perm(a, b, sel): selector values 0-3 take b's bytes, 4-7 take a's)
__device__ __forceinline__ void transpose4(const unsigned (&w)[4], unsigned (&o)[4]) {
  const unsigned a01l = __builtin_amdgcn_perm(w[1], w[0], 0x05010400u), a01h = __builtin_amdgcn_perm(w[1], w[0], 0x07030602u);
  const unsigned a23l = __builtin_amdgcn_perm(w[3], w[2], 0x05010400u), a23h = __builtin_amdgcn_perm(w[3], w[2], 0x07030602u);
  o[0] = __builtin_amdgcn_perm(a23l, a01l, 0x05040100u);
  o[1] = __builtin_amdgcn_perm(a23l, a01l, 0x07060302u);
  o[2] = __builtin_amdgcn_perm(a23h, a01h, 0x05040100u);
  o[3] = __builtin_amdgcn_perm(a23h, a01h, 0x07060302u);
}

// The digit slices of a lane's 16 K-entries (four per chunk c: the four registers of one accumulator tile) are the B operands
// of the i8 MFMAs: dig[i][c] = byte r <-> entry 4 c + r.  slice_chunk writes dword c of all seven digits from the chunk's four
// values.
__device__ __forceinline__ void slice_chunk(v4i (&dig)[kDigits], int c, const v4d& x) {
  unsigned lo[4], hi[4], tl[4], th[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) fixq(x[r], lo[r], hi[r]);
  transpose4(lo, tl);
  transpose4(hi, th);
#pragma unroll
  for (int i = 0; i < 4; ++i) dig[i][c] = (int)tl[i];
#pragma unroll
  for (int i = 4; i < kDigits; ++i) dig[i][c] = (int)th[i - 4];
}

// acc[L - kLmin] += A_i . B_j for every digit pair of level L = i + j >= kLmin, in ROUND-ROBIN order over the levels (first pair
// of every level, second pair of every level, ...): an i8 MFMA that accumulates onto the previous one's result waits ~44 clocks
// against 17 for an independent one (tools/ubench_i8emu.hip), so consecutive MFMAs must write different accumulators.
struct PairOrder {
  int i[kDigits * kDigits], j[kDigits * kDigits], n;
};
constexpr PairOrder pair_order() {
  PairOrder o{};
  o.n = 0;
  for (int round = 0; round < kDigits; ++round)
    for (int L = kLmin; L <= kTop; ++L) {
      // the round-th pair (i, L - i) of level L, i ascending
      int seen = 0;
      for (int i = 0; i < kDigits; ++i) {
        const int j = L - i;
        if (j < 0 || j >= kDigits) continue;
        if (seen == round) {
          o.i[o.n] = i;
          o.j[o.n] = j;
          ++o.n;
        }
        ++seen;
      }
    }
  return o;
}
__device__ __forceinline__ void tile_mfma(v4i (&acc)[kLevels], const v4i (&a)[kDigits], const v4i (&b)[kDigits]) {
  constexpr PairOrder o = pair_order();
#pragma unroll
  for (int k = 0; k < o.n; ++k) {
    const int i = o.i[k], j = o.j[k];
#if NLC_I8_DBG == 1  // tools only (timing): no MFMAs, the operands stay live
    asm volatile("" : "+v"(acc[i + j - kLmin]) : "v"(a[i]), "v"(b[j]));
#else
    acc[i + j - kLmin] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i + j - kLmin], 0, 0, 0);
#endif
  }
}

// pre + rs (x) sum_L 256^(L - kTop) acc[L]: Horner from the lowest level up (the small terms first), the four registers side by
// side (four independent chains: a dependent FP64 instruction waits for its producer).  MERGE: adjacent levels are first joined in
// int32, (c_(L+1) << 8) + c_L -- safe while one GEMM's digit products feed an accumulator (|c| < 2^23); an accumulator shared by
// two GEMMs (layer 1's reset / update gates: W_ih h0 + W_hh h1) is recombined level by level.  With an even number of levels the
// merged sum ends one level below the top: ITS row factors carry the extra 2^-8 (nlc_pack.h: pack_gru_i8_stream).
constexpr bool kMergedFactorShift = kLevels % 2 == 0;
template <bool MERGE>
__device__ __forceinline__ v4d recombine(const v4i (&acc)[kLevels], const v4d& rs, const v4d& pre) {
  v4d out;
#if NLC_I8_DBG == 2  // tools only (timing): the MFMAs without their recombination
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int m = acc[0][r];
#pragma unroll
    for (int l = 1; l < kLevels; ++l) asm volatile("" : "+v"(m) : "v"(acc[l][r]));
    out[r] = fma((double)m, rs[r], pre[r]);
  }
  return out;
#endif
  double s[4];
  if (MERGE) {
    constexpr int NP = (kLevels + 1) / 2;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int l = 2 * p;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = (l + 1 < kLevels) ? (int)(((unsigned)acc[l + 1][r] << 8) + (unsigned)acc[l][r]) : acc[l][r];
        s[r] = (p == 0) ? (double)m : fma(s[r], 0x1p-16, (double)m);
      }
      // (left alone the compiler lays the four chains out one after the other, cvt, fma, cvt, fma ... back to back; tying the four
      // partial sums to one empty asm keeps them side by side)
      asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]));
    }
  } else {
#pragma unroll
    for (int l = 0; l < kLevels; ++l) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] = (l == 0) ? (double)acc[l][r] : fma(s[r], 0x1p-8, (double)acc[l][r]);
      asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]));
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) out[r] = fma(s[r], rs[r], pre[r]);
  return out;
}
// digit fragments of one 16-row weight tile: frag[i] = 16 bytes per lane, [digit i][lane][16]
__device__ __forceinline__ void load_tile(v4i (&a)[kDigits], const signed char* __restrict__ tile, int lane) {
  typedef const __attribute__((address_space(1))) v4i* g4;
  g4 p = (g4)tile;
  asm volatile("" : "+s"(p));
#pragma unroll
  for (int i = 0; i < kDigits; ++i) a[i] = p[i * 64 + lane];
}

}  // namespace i8
}  // namespace nlc
