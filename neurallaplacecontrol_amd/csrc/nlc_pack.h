// Host-side weight re-packing for the MFMA kernels: pure C++ (no HIP), shared by abi_model.hip and the CPU
// emulation test (tests/helpers/pack_host.cpp), which replays the kernels' dataflow lane by lane.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>
#include <utility>
#include <vector>

namespace nlc {

// ---- MFMA A-fragment packing:  out[(ks*MT + mt)*64 + lane] = W[row(16 mt + (lane & 15))][4 ks + (lane >> 4)]
// rowmap[i] = source row of packed row i (-1 = zero row); cols beyond K are zero.
inline std::vector<double> pack_A(const double* W, int ldw, int K, const std::vector<int>& rowmap) {
  const int MT = (int)(rowmap.size() + 15) / 16;
  const int KS = (K + 3) / 4;
  std::vector<double> out((size_t)KS * MT * 64, 0.0);
  for (int ks = 0; ks < KS; ++ks)
    for (int mt = 0; mt < MT; ++mt)
      for (int lane = 0; lane < 64; ++lane) {
        const int prow = 16 * mt + (lane & 15), col = 4 * ks + (lane >> 4);
        if (prow >= (int)rowmap.size() || col >= K) continue;
        const int src = rowmap[prow];
        if (src < 0) continue;
        out[((size_t)ks * MT + mt) * 64 + lane] = W[(size_t)src * ldw + col];
      }
  return out;
}
// GRU gate matrices, chunk-packed: out[((j*KS + ks)*3 + g)*64 + lane] = W[g*G + 16 j + (lane & 15)][4 ks + (lane >> 4)]
inline std::vector<double> pack_gru_chunked(const double* W, int ldw, int K, int G) {
  const int GT = G / 16, KS = (K + 3) / 4;
  std::vector<double> out((size_t)GT * KS * 3 * 64, 0.0);
  for (int j = 0; j < GT; ++j)
    for (int ks = 0; ks < KS; ++ks)
      for (int g = 0; g < 3; ++g)
        for (int lane = 0; lane < 64; ++lane) {
          const int row = g * G + 16 * j + (lane & 15), col = 4 * ks + (lane >> 4);
          if (col >= K) continue;
          out[(((size_t)j * KS + ks) * 3 + g) * 64 + lane] = W[(size_t)row * ldw + col];
        }
  return out;
}
inline std::vector<int> identity_rows(int n) {
  std::vector<int> r(n);
  for (int i = 0; i < n; ++i) r[i] = i;
  return r;
}


// ---- layer-3 slot layout of the fused sphere-map / Fourier-ILT epilogue (kernels_nl.hip).
// Elements (dim c, term k): all even-k pairs, padded to a multiple of 4, then all odd-k pairs (scale == 2: the
// Fourier phase is i^k, even terms need cos(theta), odd terms sin(theta)).  Group g = 4 consecutive elements =
// the four lane groups q of one accumulator register; tile j, register r in {0,1} holds theta of group 2j+r,
// register r+2 its phi.
struct IltSlots {
  int nt3 = 0, n_even_groups = 0;
  std::vector<std::pair<int, int>> elems;  // (dim, k) per slot, (-1,-1) = padding; size 8*nt3
  std::vector<int> rowmap3;                // packed row -> source row of the last Linear (or -1); size 16*nt3
  std::vector<double> Cp;                  // coefficient fragments [2*nt3][64]
};
inline int ilt_tiles_needed(int d, int S) {
  const int n_even = d * ((S + 1) / 2), n_odd = d * (S / 2);
  const int groups = (n_even + 3) / 4 + (n_odd + 3) / 4;
  return (groups + 1) / 2;
}
inline IltSlots make_ilt_slots(int d, int S, int nt3) {
  IltSlots s;
  s.nt3 = nt3;
  for (int c = 0; c < d; ++c)
    for (int k = 0; k < S; k += 2) s.elems.emplace_back(c, k);
  while (s.elems.size() % 4) s.elems.emplace_back(-1, -1);
  s.n_even_groups = (int)s.elems.size() / 4;
  for (int c = 0; c < d; ++c)
    for (int k = 1; k < S; k += 2) s.elems.emplace_back(c, k);
  s.elems.resize((size_t)nt3 * 8, {-1, -1});
  s.rowmap3.assign((size_t)nt3 * 16, -1);
  for (int j = 0; j < nt3; ++j)
    for (int m = 0; m < 16; ++m) {
      const int r = m >> 2, q = m & 3;
      const auto el = s.elems[(size_t)4 * (2 * j + (r & 1)) + q];
      if (el.first < 0) continue;
      s.rowmap3[(size_t)16 * j + m] = (r < 2 ? 0 : d * S) + el.first * S + el.second;
    }
  // C[dim][element] = w_k * Re-part selector of i^k (k even: cos, +1/-1; k odd: sin, -1/+1); fragment layout of an
  // MFMA A operand: lane -> (row m = lane & 15 = dim, k-index lane >> 4 = slot within the group)
  s.Cp.assign((size_t)2 * nt3 * 64, 0.0);
  for (int g = 0; g < 2 * nt3; ++g)
    for (int lane = 0; lane < 64; ++lane) {
      const auto el = s.elems[(size_t)4 * g + (lane >> 4)];
      if (el.first != (lane & 15)) continue;
      const int k = el.second;
      s.Cp[(size_t)g * 64 + lane] = ((k == 0) ? 0.5 : 1.0) * (((k & 3) == 0 || (k & 3) == 3) ? 1.0 : -1.0);
    }
  return s;
}

// ---- int8 digit fragments of a GRU gate matrix for the INT8 matrix pipe (nlc_i8gemm.h: FP64 GEMMs with bounded operands as
// sliced fixed-point products).  Row m of the matrix is scaled by 2^-e_m into [-1, 1] (e_m from rowexp: the caller makes two
// matrices that feed ONE accumulator share their exponents), written as X = rint(w 2^(54 - e_m)) and cut into seven signed
// digits, X = sum_i d_i 256^i (d_0 .. d_5 in [-128, 127] by the 128-per-digit bias, d_6 the signed rest).
constexpr int kI8Digits = 7;
constexpr int kI8Frac = 54;                          // nlc_i8gemm.h: kFrac
#ifndef NLC_I8_LMIN
#define NLC_I8_LMIN 5
#endif
constexpr int kI8Lmin = NLC_I8_LMIN;                 // nlc_i8gemm.h: kLmin
constexpr int kI8RowExp2 = 8 * 12 - 2 * kI8Frac;     // nlc_i8gemm.h: kRowExp2 (the recombined sum is in units of 256^12 2^-108)
inline void i8_digits(double x, signed char d[kI8Digits]) {  // |x| <= 1
  const long long X = std::llrint(std::ldexp(x, kI8Frac));
  const unsigned long long Xb = (unsigned long long)X + 0x0000808080808080ull;
  for (int i = 0; i < 6; ++i) d[i] = (signed char)(unsigned char)(((Xb >> (8 * i)) & 0xff) ^ 0x80);
  d[6] = (signed char)(unsigned char)((Xb >> 48) & 0xff);
}
// smallest e with max |row| <= 2^e (0 for a zero row)
inline std::vector<int> i8_row_exponents(const double* W, int rows, int K) {
  std::vector<int> e((size_t)rows, 0);
  for (int r = 0; r < rows; ++r) {
    double mx = 0.0;
    for (int k = 0; k < K; ++k) mx = std::fmax(mx, std::fabs(W[(size_t)r * K + k]));
    if (mx > 0.0) {
      int ex;
      const double f = std::frexp(mx, &ex);  // mx = f 2^ex, f in [0.5, 1)
      e[r] = (f == 0.5) ? ex - 1 : ex;       // mx == 2^(ex - 1) exactly: |w| / 2^(ex - 1) <= 1 holds already
    }
  }
  return e;
}
// Fragments of the (3 G) x G gate matrix W (K = G = 64: ONE v_mfma_i32_16x16x64_i8 per digit pair):
//   out[(((j * 3 + g) * 7 + i) * 64 + lane) * 16 + b],  chunk j, gate g, digit i
//   lane: output row m = lane & 15 of the tile -- the i32 accumulator holds row 4 q + r in register r of lane group q, and the
//         kernels keep feature 16 j + 4 r + q there (the FP64 MFMA's layout), so tile row m is feature 16 j + 4 (m & 3) + (m >> 2);
//   byte b of lane group kq = lane >> 4: K entry 16 (b >> 2) + 4 (b & 3) + kq -- the entry the B operand's lane group kq holds
//         in byte b (dword = chunk, byte = accumulator register: i8::slice_chunk).
inline std::vector<signed char> pack_gru_i8(const double* W, int G, const std::vector<int>& rowexp) {
  const int GT = G / 16;
  std::vector<signed char> out((size_t)GT * 3 * kI8Digits * 64 * 16, 0);
  for (int j = 0; j < GT; ++j)
    for (int g = 0; g < 3; ++g)
      for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 15, kq = lane >> 4;
        const int row = g * G + 16 * j + 4 * (m & 3) + (m >> 2);
        for (int b = 0; b < 16; ++b) {
          const int k = 16 * (b >> 2) + 4 * (b & 3) + kq;
          signed char d[kI8Digits];
          i8_digits(std::ldexp(W[(size_t)row * G + k], -rowexp[row]), d);
          for (int i = 0; i < kI8Digits; ++i) out[((((size_t)j * 3 + g) * kI8Digits + i) * 64 + lane) * 16 + b] = d[i];
        }
      }
  return out;
}
inline std::vector<double> i8_row_factors(const std::vector<int>& rowexp) {
  std::vector<double> f(rowexp.size());
  for (size_t r = 0; r < rowexp.size(); ++r) f[r] = std::ldexp(1.0, rowexp[r] + kI8RowExp2);
  return f;
}

// The encoder's weight stream (kernels_gru_i8.hip): the 36 gate tiles of one GRU step in the order the kernel consumes them --
//   blocks 0 .. 11   layer 0, W_hh:   chunk j, gate g (r, z, n)                         -> 3 j + g
//   blocks 12 .. 35  layer 1:         chunk j, gate g: W_hh tile, then W_ih tile        -> 12 + 6 j + 2 g (+ 1)
//                    (hidden side first: the tile that opens layer 1 must not need the new h0, whose digits are cut in its shadow)
// -- each block = the tile's seven digit fragments (7 168 B), its 16 recombination factors and its 16 biases (doubles, feature
// order f = 0 .. 15 of the tile).  Biases: layer 0's reset / update biases ride in the input GEMM (none here), its n tile carries
// b_hn; layer 1's reset / update tiles BOTH carry b_ih + b_hh (one accumulator, recombined after the second tile; step 0 has no
// second tile), its n tiles b_in and b_hn.  Reset / update rows of layer 1 share their exponents between W_ih and W_hh.
constexpr int kI8StreamBlocks = 36;
constexpr int kI8BlockBytes = kI8Digits * 64 * 16 + 2 * 16 * 8;
inline std::vector<signed char> pack_gru_i8_stream(const double* Whh0, const double* Wih1, const double* Whh1, const double* bhn0,
                                                   const double* brz1, const double* bin1, const double* bhn1, int G) {
  const int GT = G / 16;
  const std::vector<int> e_hh0 = i8_row_exponents(Whh0, 3 * G, G);
  std::vector<int> e_ih1 = i8_row_exponents(Wih1, 3 * G, G), e_hh1 = i8_row_exponents(Whh1, 3 * G, G);
  for (int r = 0; r < 2 * G; ++r) e_ih1[r] = e_hh1[r] = (e_ih1[r] > e_hh1[r] ? e_ih1[r] : e_hh1[r]);
  const std::vector<signed char> f_hh0 = pack_gru_i8(Whh0, G, e_hh0), f_ih1 = pack_gru_i8(Wih1, G, e_ih1), f_hh1 = pack_gru_i8(Whh1, G, e_hh1);
  const std::vector<double> r_hh0 = i8_row_factors(e_hh0), r_ih1 = i8_row_factors(e_ih1), r_hh1 = i8_row_factors(e_hh1);
  std::vector<signed char> out((size_t)kI8StreamBlocks * kI8BlockBytes, 0);
  const size_t frag = (size_t)kI8Digits * 64 * 16;
  // (i8::recombine<MERGE = true> with an even level count ends one level below the top: the tiles it serves -- every tile but
  // layer 1's reset / update pairs -- carry their factors times 2^-8)
  constexpr int kLevels = 2 * (kI8Digits - 1) - kI8Lmin + 1;
  auto block = [&](int t, const std::vector<signed char>& f, const std::vector<double>& rf, int j, int g, const double* bias /*16 or null*/,
                   bool merged) {
    signed char* b = out.data() + (size_t)t * kI8BlockBytes;
    std::memcpy(b, f.data() + ((size_t)j * 3 + g) * frag, frag);
    const double shift = (merged && kLevels % 2 == 0) ? 0x1p-8 : 1.0;
    double tail[32];
    for (int i = 0; i < 16; ++i) {
      tail[i] = rf[(size_t)g * G + 16 * j + i] * shift;
      tail[16 + i] = bias ? bias[i] : 0.0;
    }
    std::memcpy(b + frag, tail, sizeof(tail));
  };
  for (int j = 0; j < GT; ++j)
    for (int g = 0; g < 3; ++g) {
      block(3 * j + g, f_hh0, r_hh0, j, g, g == 2 ? bhn0 + 16 * j : nullptr, true);
      const double* b_ih = g < 2 ? brz1 + g * G + 16 * j : bin1 + 16 * j;
      const double* b_hh = g < 2 ? brz1 + g * G + 16 * j : bhn1 + 16 * j;
      block(12 + 6 * j + 2 * g, f_hh1, r_hh1, j, g, b_hh, g == 2);
      block(12 + 6 * j + 2 * g + 1, f_ih1, r_ih1, j, g, b_ih, g == 2);
    }
  return out;
}

}  // namespace nlc
