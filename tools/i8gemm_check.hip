// Unit check of csrc/nlc_i8gemm.h on the GPU (gfx950), against exact host arithmetic:
//   1. fixq / slice_chunk: the seven digits of rint(x 2^54) for random and edge values of x in [-1, 1];
//   2. v_mfma_i32_16x16x64_i8: which K entry a byte of the A / B operand is, which output entry an accumulator register is;
//   3. a whole sliced GEMM tile (16 rows x K = 64 x 16 columns): error against the exact product (__int128 / long double), beside
//      the error of the FP64 fused-multiply-add chain the native MFMA path evaluates.
// hipcc --offload-arch=gfx950 -O3 -I neurallaplacecontrol_amd/csrc tools/i8gemm_check.hip -o tools/i8gemm_check.bin && tools/i8gemm_check.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <random>
#include <vector>

#include "nlc_i8gemm.h"
#include "nlc_pack.h"

using namespace nlc;
using nlc::i8::v4i;

__global__ void k_slices(const double* x /*[64 lanes][16]*/, int* out /*[64][7][4]*/) {
  const int lane = threadIdx.x;
  v4i dig[i8::kDigits];
  for (int c = 0; c < 4; ++c) {
    v4d v = {x[lane * 16 + 4 * c + 0], x[lane * 16 + 4 * c + 1], x[lane * 16 + 4 * c + 2], x[lane * 16 + 4 * c + 3]};
    if (c == 0) i8::slice_chunk(dig, 0, v);
    if (c == 1) i8::slice_chunk(dig, 1, v);
    if (c == 2) i8::slice_chunk(dig, 2, v);
    if (c == 3) i8::slice_chunk(dig, 3, v);
  }
  for (int i = 0; i < i8::kDigits; ++i)
    for (int c = 0; c < 4; ++c) out[(lane * i8::kDigits + i) * 4 + c] = dig[i][c];
}

__global__ void k_mfma(const v4i* a, const v4i* b, v4i* d) {
  const int lane = threadIdx.x;
  v4i acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[lane], b[lane], acc, 0, 0, 0);
  d[lane] = acc;
}

// one sliced GEMM tile: h as the kernels hold it (lane (q, n), chunk c, register r <-> feature 16 c + 4 r + q of column n)
template <bool MERGE>
__global__ void k_gemm(const signed char* wfrag, const double* rowfac /*16, tile row order of the accumulator*/, const double* h /*[64][16]*/,
                       double* out /*[64 lanes][4]*/) {
  const int lane = threadIdx.x, q = lane >> 4, n = lane & 15;
  v4i dig[i8::kDigits];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    v4d v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = h[(16 * c + 4 * r + q) * 16 + n];
    i8::slice_chunk(dig, c, v);
  }
  v4i a[i8::kDigits];
  i8::load_tile(a, wfrag, lane);
  v4i acc[i8::kLevels];
#pragma unroll
  for (int l = 0; l < i8::kLevels; ++l) acc[l] = v4i{0, 0, 0, 0};
  i8::tile_mfma(acc, a, dig);
  const double shift = (MERGE && i8::kMergedFactorShift) ? 0x1p-8 : 1.0;  // (the stream's blocks carry it: pack_gru_i8_stream)
  const v4d rs = {rowfac[q + 0] * shift, rowfac[q + 4] * shift, rowfac[q + 8] * shift, rowfac[q + 12] * shift};  // register r <-> feature 4 r + q
  const v4d res = i8::recombine<MERGE>(acc, rs, v4d{0, 0, 0, 0});
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = res[r];
}

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 2;                                                                \
    }                                                                          \
  } while (0)

// ---------------------------------------------------------------------------------------------------------------------------
// --adversarial FILE: sliced GEMM tiles on rows built to break a fixed-point product, dumped for tests/test_gpu_i8_gemm.py, which
// checks every output against EXACT rational arithmetic (python fractions).  Record per tile: int32 family, 16 x 64 W (rows of the
// tile), 16 row scales s_m = 2^e, 64 x 16 h ([k][column]), 16 x 16 outputs with merged levels, 16 x 16 level by level (row-major
// [row][column]).  Families:
//   0 cancelling rows (sum w h ~ 0, sum |w h| large)      1 a 2^+-40 spread of |w| and of |h| inside a row, large against small
//   2 h = +-1 exactly, |w| up to the row scale exactly     3 every |h| < 2^-54 (quantised to zero)   4 a few tiny h among ordinary ones
//   5 ordinary random draws (the control)
static int run_adversarial(const char* path) {
  std::mt19937_64 rng(2026);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  const int G = 64, per_family = 8, families = 6;
  FILE* f = fopen(path, "wb");
  if (!f) return 2;
  const int n_tiles = per_family * families;
  fwrite(&n_tiles, 4, 1, f);
  signed char* dw;
  double *drf, *dh, *dout;
  CK(hipMalloc(&dw, 7 * 64 * 16));
  CK(hipMalloc(&drf, 16 * 8));
  CK(hipMalloc(&dh, 64 * 16 * 8));
  CK(hipMalloc(&dout, 64 * 4 * 8));
  for (int fam = 0; fam < families; ++fam)
    for (int t = 0; t < per_family; ++t) {
      std::vector<double> W((size_t)3 * G * G), h(64 * 16);
      for (auto& v : W) v = U(rng);
      for (auto& v : h) v = U(rng);
      const double span = std::ldexp(1.0, (int)(rng() % 7) - 3);  // row scales 2^-3 .. 2^3
      for (int row = 0; row < 16; ++row)
        for (int k = 0; k < G; ++k) W[(size_t)row * G + k] *= span;
      if (fam == 0) {
        // pairs (2 i, 2 i + 1) whose products cancel to the last bits: h_(2i+1) = -(w_2i h_2i) / w_(2i+1), for ONE row per column
        for (int n = 0; n < 16; ++n) {
          const int row = n;
          for (int i = 0; i < G / 2; ++i) {
            double &w0 = W[(size_t)row * G + 2 * i], &w1 = W[(size_t)row * G + 2 * i + 1];
            if (std::fabs(w1) < std::fabs(w0)) std::swap(w0, w1);
            if (w1 == 0.0) w1 = span;
            h[(2 * i + 1) * 16 + n] = -(w0 * h[(2 * i) * 16 + n]) / w1;
          }
        }
      } else if (fam == 1) {
        for (int row = 0; row < 16; ++row)
          for (int k = 0; k < G; ++k) {
            const bool big = ((k + row) % 2) == 0;
            W[(size_t)row * G + k] = std::ldexp(W[(size_t)row * G + k], big ? 0 : -40 + (int)(rng() % 9));
          }
        for (int k = 0; k < G; ++k)
          for (int n = 0; n < 16; ++n) {
            const bool big_w = ((k + n) % 2) == 0;  // row n: large weights meet small states and the other way round
            h[k * 16 + n] = std::ldexp(h[k * 16 + n], big_w ? -40 + (int)(rng() % 9) : 0);
          }
      } else if (fam == 2) {
        for (auto& v : h) v = (rng() & 1) ? 1.0 : -1.0;
        for (int row = 0; row < 16; ++row)
          for (int k = 0; k < G; k += 5) W[(size_t)row * G + k] = ((rng() & 1) ? 1.0 : -1.0) * span;  // |w| = 2^e exactly
      } else if (fam == 3) {
        for (auto& v : h) v = std::ldexp(v, -55 - (int)(rng() % 40));
        if (t == 0) h[5] = 0x1p-1074;
      } else if (fam == 4) {
        for (size_t i = 0; i < h.size(); i += 3) h[i] = std::ldexp(h[i], -55 - (int)(rng() % 900));
      }
      const auto rexp = i8_row_exponents(W.data(), 3 * G, G);
      const auto frag = pack_gru_i8(W.data(), G, rexp);
      const auto rf = i8_row_factors(rexp);
      CK(hipMemcpy(dw, frag.data(), 7 * 64 * 16, hipMemcpyHostToDevice));
      CK(hipMemcpy(drf, rf.data(), 16 * 8, hipMemcpyHostToDevice));
      CK(hipMemcpy(dh, h.data(), h.size() * 8, hipMemcpyHostToDevice));
      fwrite(&fam, 4, 1, f);
      fwrite(W.data(), 8, 16 * G, f);
      double scale[16];
      for (int r = 0; r < 16; ++r) scale[r] = std::ldexp(1.0, rexp[r]);
      fwrite(scale, 8, 16, f);
      fwrite(h.data(), 8, h.size(), f);
      for (int merge = 1; merge >= 0; --merge) {
        if (merge) hipLaunchKernelGGL(k_gemm<true>, dim3(1), dim3(64), 0, 0, dw, drf, dh, dout);
        else hipLaunchKernelGGL(k_gemm<false>, dim3(1), dim3(64), 0, 0, dw, drf, dh, dout);
        std::vector<double> out(64 * 4), tile(256);
        CK(hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost));
        for (int lane = 0; lane < 64; ++lane)
          for (int r = 0; r < 4; ++r) tile[(4 * r + (lane >> 4)) * 16 + (lane & 15)] = out[lane * 4 + r];
        fwrite(tile.data(), 8, 256, f);
      }
    }
  fclose(f);
  printf("adversarial tiles: %d written to %s\n", n_tiles, path);
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 2 && std::strcmp(argv[1], "--adversarial") == 0) return run_adversarial(argv[2]);
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  int bad = 0;
  // ------------------------------------------------------------------ 1. digits
  {
    std::vector<double> x(64 * 16);
    for (auto& v : x) v = U(rng);
    const double edge[] = {0.0, 1.0, -1.0, 0x1p-52, -0x1p-52, 0x1p-53, 0x1.8p-53, -0x1.8p-53, 1.0 - 0x1p-53, -1.0 + 0x1p-53, 0.5, -0.5,
                           0x1p-30, -0x1p-30, 0x1.fffffffffffffp-31, 0x1p-22, 0x1.8p-23, -0x1.8p-23, 0x1p-1074, 1e-300, 0.999999, -0.999999,
                           0x1.0000000000001p-1, 0x1.123456789abcdp-3, -0x1.fedcba9876543p-7, 0x1p-23 + 0x1p-53, 0x1.8p-22 + 0x1.8p-52};
    for (size_t i = 0; i < sizeof(edge) / sizeof(edge[0]); ++i) x[i * 7 % x.size()] = edge[i];
    double* dx;
    int* dout;
    CK(hipMalloc(&dx, x.size() * 8));
    CK(hipMalloc(&dout, 64 * 7 * 4 * 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_slices, dim3(1), dim3(64), 0, 0, dx, dout);
    std::vector<int> out(64 * 7 * 4);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    int wrong = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
          signed char d[7];
          i8_digits(x[lane * 16 + 4 * c + r], d);
          for (int i = 0; i < 7; ++i) {
            const signed char got = (signed char)((out[(lane * 7 + i) * 4 + c] >> (8 * r)) & 0xff);
            if (got != d[i]) {
              if (wrong < 8) printf("  digit %d of x = %a: device %d, host %d\n", i, x[lane * 16 + 4 * c + r], got, d[i]);
              ++wrong;
            }
          }
        }
    printf("1. digits of rint(x 2^54), %zu values: %d wrong bytes\n", x.size(), wrong);
    bad += wrong;
  }
  // ------------------------------------------------------------------ 2. operand / accumulator layout of the i8 MFMA
  {
    std::vector<signed char> A(16 * 64), B(64 * 16);  // A[m][k], B[k][n], k = 16 kq + byte
    for (auto& v : A) v = (signed char)(rng() % 256);
    for (auto& v : B) v = (signed char)(rng() % 256);
    std::vector<int> af(64 * 4), bf(64 * 4);
    for (int lane = 0; lane < 64; ++lane)
      for (int b = 0; b < 16; ++b) {
        const int k = 16 * (lane >> 4) + b;
        ((signed char*)af.data())[lane * 16 + b] = A[(lane & 15) * 64 + k];
        ((signed char*)bf.data())[lane * 16 + b] = B[k * 16 + (lane & 15)];
      }
    v4i *da, *db, *dd;
    CK(hipMalloc(&da, 64 * 16));
    CK(hipMalloc(&db, 64 * 16));
    CK(hipMalloc(&dd, 64 * 16));
    CK(hipMemcpy(da, af.data(), 64 * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, bf.data(), 64 * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dd);
    std::vector<int> D(64 * 4);
    CK(hipMemcpy(D.data(), dd, 64 * 16, hipMemcpyDeviceToHost));
    int wrong_a = 0, wrong_b = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int r = 0; r < 4; ++r) {
        const int q = lane >> 4, n = lane & 15;
        long ref_a = 0, ref_b = 0;
        for (int k = 0; k < 64; ++k) {
          ref_a += (long)A[(4 * q + r) * 64 + k] * B[k * 16 + n];  // row 4 q + r
          ref_b += (long)A[(q + 4 * r) * 64 + k] * B[k * 16 + n];  // row q + 4 r (the FP64 MFMA's layout)
        }
        wrong_a += D[lane * 4 + r] != ref_a;
        wrong_b += D[lane * 4 + r] != ref_b;
      }
    printf("2. v_mfma_i32_16x16x64_i8 with byte b of lane group kq = K entry 16 kq + b on both operands: accumulator register r of lane\n"
           "   group q = row 4 q + r: %d wrong of 256; = row q + 4 r: %d wrong of 256\n", wrong_a, wrong_b);
    bad += wrong_a;
  }
  // ------------------------------------------------------------------ 3. sliced GEMM tiles against the exact product
  {
    const int G = 64, trials = 200;
    double worst_i8 = 0, worst_i8_plain = 0, worst_f64 = 0, sum_i8 = 0, sum_f64 = 0;
    long count = 0;
    signed char* dw;
    double *drf, *dh, *dout;
    CK(hipMalloc(&dw, 7 * 64 * 16));
    CK(hipMalloc(&drf, 16 * 8));
    CK(hipMalloc(&dh, 64 * 16 * 8));
    CK(hipMalloc(&dout, 64 * 4 * 8));
    for (int t = 0; t < trials; ++t) {
      // a 3G x G matrix whose first tile we use; magnitudes spread over rows and entries, as trained weights are
      std::vector<double> W((size_t)3 * G * G);
      const double span = (t % 4 == 0) ? 1.0 : (t % 4 == 1 ? 1e-3 : (t % 4 == 2 ? 30.0 : 0.3));
      for (size_t i = 0; i < W.size(); ++i) W[i] = U(rng) * span * ((rng() % 8 == 0) ? 1e-4 : 1.0);
      std::vector<double> h(64 * 16);
      for (auto& v : h) v = (t % 5 == 4) ? (U(rng) > 0 ? 1.0 : -1.0) * (1.0 - std::fabs(U(rng)) * 1e-6) : U(rng) * ((t % 3 == 0) ? 1.0 : 0.05);
      const auto rexp = i8_row_exponents(W.data(), 3 * G, G);
      const auto frag = pack_gru_i8(W.data(), G, rexp);
      const auto rf = i8_row_factors(rexp);
      CK(hipMemcpy(dw, frag.data(), 7 * 64 * 16, hipMemcpyHostToDevice));  // chunk 0, gate 0
      CK(hipMemcpy(drf, rf.data(), 16 * 8, hipMemcpyHostToDevice));        // rows 0 .. 15
      CK(hipMemcpy(dh, h.data(), h.size() * 8, hipMemcpyHostToDevice));
      for (int merge = 0; merge < 2; ++merge) {
        if (merge) hipLaunchKernelGGL(k_gemm<true>, dim3(1), dim3(64), 0, 0, dw, drf, dh, dout);
        else hipLaunchKernelGGL(k_gemm<false>, dim3(1), dim3(64), 0, 0, dw, drf, dh, dout);
        std::vector<double> out(64 * 4);
        CK(hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost));
        for (int lane = 0; lane < 64; ++lane)
          for (int r = 0; r < 4; ++r) {
            const int q = lane >> 4, n = lane & 15, row = 4 * r + q;
            long double exact = 0, mag = 0;
            double chain = 0;
            for (int k = 0; k < G; ++k) {
              exact += (long double)W[(size_t)row * G + k] * (long double)h[k * 16 + n];
              mag += fabsl((long double)W[(size_t)row * G + k] * (long double)h[k * 16 + n]);
              chain = std::fma(W[(size_t)row * G + k], h[k * 16 + n], chain);  // the FP64 MFMA path: 16 k-steps of 4, one rounding each
            }
            // errors in units of 2^-53 times the row's sum of |w h|
            const double u = (double)(mag > 0 ? mag : 1) * 0x1p-53;
            const double e_i8 = std::fabs((double)((long double)out[lane * 4 + r] - exact)) / u;
            const double e_f64 = std::fabs((double)((long double)chain - exact)) / u;
            if (merge) {
              worst_i8 = std::fmax(worst_i8, e_i8);
              sum_i8 += e_i8;
              worst_f64 = std::fmax(worst_f64, e_f64);
              sum_f64 += e_f64;
              ++count;
            } else {
              worst_i8_plain = std::fmax(worst_i8_plain, e_i8);
            }
          }
      }
    }
    printf("3. sliced GEMM tile (K = 64, kLmin = %d, %d levels), %ld outputs over %d random weight / state draws; error against the exact product\n"
           "   in units of 2^-53 sum_k |w h|:  int8-sliced (merged levels) max %.2f mean %.3f | (level by level) max %.2f | FP64 fma chain max %.2f mean %.3f\n",
           i8::kLmin, i8::kLevels, count, trials, worst_i8, sum_i8 / count, worst_i8_plain, worst_f64, sum_f64 / count);
    if (!(worst_i8 < 64.0) || !(worst_i8_plain < 64.0)) ++bad;  // the bound the header states: a few 2^-52 of the row's |w||h| sum
  }
  printf(bad ? "FAILED\n" : "OK\n");
  return bad ? 1 : 0;
}
