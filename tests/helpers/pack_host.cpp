// CPU emulation of the MFMA kernels' dataflow (tests/test_pack_emulation.py, g++, no GPU): replays, lane by lane,
// what kernels_gru.hip / kernels_nl.hip do with the fragment-packed weights produced by csrc/nlc_pack.h, using the
// v_mfma_f64_16x16x4_f64 operand maps documented in csrc/nlc_device.h.  A wrong packing order, slot permutation,
// coefficient matrix or accumulator-as-next-operand assumption shows up here, before any GPU run.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../neurallaplacecontrol_amd/csrc/nlc_pack.h"

using namespace nlc;
typedef double Acc[64][4];

// D(16x16) += A(16x4) B(4x16); A: lane l -> [m=l&15][k=l>>4]; B: lane l -> [k=l>>4][n=l&15];
// D: lane l, reg r -> [row=(l>>4)+4r][col=l&15]
static void mfma(const double* a, const double* b, Acc acc) {
  double A[16][4], B[4][16];
  for (int l = 0; l < 64; ++l) {
    A[l & 15][l >> 4] = a[l];
    B[l >> 4][l & 15] = b[l];
  }
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int row = (l >> 4) + 4 * r, col = l & 15;
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += A[row][k] * B[k][col];
      acc[l][r] += s;
    }
}
static void bias_tile(const double* b, int j, Acc acc) {
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) acc[l][r] = b[16 * j + 4 * r + (l >> 4)];
}

extern "C" {

// representation MLP + sphere map + Fourier ILT for 16 samples.  p: (16, P) latent rows [obs_n | pa]; out: (16, d)
void emu_mlp_ilt(const double* W1, const double* b1, const double* W2, const double* b2, const double* W3,
                 const double* b3, int d, int S, int h, const double* p, double tn, double alpha, double log_tol,
                 double scale, double* out) {
  const int P = d + 2, HT = h / 16, KS = h / 4;
  // constant sphere inputs folded into the layer-1 bias (nlc_mppi_configure)
  const double Tt = scale * tn, gamma = alpha - log_tol / (scale * Tt);
  std::vector<double> sph(2 * S), bf(h), W1p((size_t)h * 8, 0.0);
  for (int k = 0; k < S; ++k) {
    const double im = M_PI * k / Tt, a2 = gamma * gamma + im * im;
    sph[k] = std::atan2(im, gamma);
    sph[S + k] = std::asin((a2 - 1.0) / (a2 + 1.0));
  }
  for (int r = 0; r < h; ++r) {
    double acc = b1[r];
    for (int j = 0; j < 2 * S; ++j) acc += W1[(size_t)r * (2 * S + P) + j] * sph[j];
    bf[r] = acc;
    for (int j = 0; j < P; ++j) W1p[(size_t)r * 8 + j] = W1[(size_t)r * (2 * S + P) + 2 * S + j];
  }
  const auto rowsh = identity_rows(h);
  const auto W1pp = pack_A(W1p.data(), 8, 8, rowsh);
  const auto W2p = pack_A(W2, h, h, rowsh);
  const int nt3 = ilt_tiles_needed(d, S);
  const IltSlots sl = make_ilt_slots(d, S, nt3);
  const auto W3p = pack_A(W3, h, h, sl.rowmap3);
  std::vector<double> b3p((size_t)nt3 * 16, 0.0);
  for (size_t i = 0; i < sl.rowmap3.size(); ++i)
    if (sl.rowmap3[i] >= 0) b3p[i] = b3[sl.rowmap3[i]];

  double p0[64], p1[64];
  for (int l = 0; l < 64; ++l) {
    const int q = l >> 4, c = l & 15;
    p0[l] = q < P ? p[c * P + q] : 0.0;
    p1[l] = 4 + q < P ? p[c * P + 4 + q] : 0.0;
  }
  std::vector<Acc> h1(HT), h2(HT), o(nt3);
  for (int j = 0; j < HT; ++j) bias_tile(bf.data(), j, h1[j]);
  for (int ks = 0; ks < 2; ++ks)
    for (int m = 0; m < HT; ++m) mfma(&W1pp[((size_t)ks * HT + m) * 64], ks ? p1 : p0, h1[m]);
  for (int j = 0; j < HT; ++j)
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) h1[j][l][r] = std::tanh(h1[j][l][r]);
  auto bfrag = [](std::vector<Acc>& hh, int ks, double* b) {
    for (int l = 0; l < 64; ++l) b[l] = hh[ks >> 2][l][ks & 3];  // accumulator register == next B fragment
  };
  double b[64];
  for (int j = 0; j < HT; ++j) bias_tile(b2, j, h2[j]);
  for (int ks = 0; ks < KS; ++ks) {
    bfrag(h1, ks, b);
    for (int m = 0; m < HT; ++m) mfma(&W2p[((size_t)ks * HT + m) * 64], b, h2[m]);
  }
  for (int j = 0; j < HT; ++j)
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) h2[j][l][r] = std::tanh(h2[j][l][r]);
  for (int j = 0; j < nt3; ++j) bias_tile(b3p.data(), j, o[j]);
  for (int ks = 0; ks < KS; ++ks) {
    bfrag(h2, ks, b);
    for (int m = 0; m < nt3; ++m) mfma(&W3p[((size_t)ks * nt3 + m) * 64], b, o[m]);
  }
  Acc ax;
  std::memset(ax, 0, sizeof(ax));
  for (int j = 0; j < nt3; ++j)
    for (int r = 0; r < 2; ++r) {
      const int g = 2 * j + r;
      double val[64];
      for (int l = 0; l < 64; ++l) {
        const double theta = std::tanh(o[j][l][r]) * M_PI;
        const double phi = std::tanh(o[j][l][r + 2]) * M_PI / 2.0 - M_PI / 2.0 + M_PI / 2.0;
        const double rad = std::tan(phi / 2.0 + M_PI / 4.0);
        val[l] = rad * (g < sl.n_even_groups ? std::cos(theta) : std::sin(theta));
      }
      mfma(&sl.Cp[(size_t)g * 64], val, ax);
    }
  const double factor = std::exp(gamma * tn) / Tt;
  for (int l = 0; l < 64; ++l) {
    const int q = l >> 4, c = l & 15;
    if (q < d) out[c * d + q] = factor * ax[l][0];
    if (4 + q < d) out[c * d + 4 + q] = factor * ax[l][1];
  }
}

// GRU encoder on chunk-packed weights, 16 windows.  win: (16, B, nin) NORMALISED; out: (16, 2)
void emu_gru(const double* Wih0, const double* Whh0, const double* bih0, const double* bhh0, const double* Wih1,
             const double* Whh1, const double* bih1, const double* bhh1, const double* Wo, const double* bo, int g,
             int nin, int B, const double* win, double* out) {
  const int GT = g / 16, KS = g / 4;
  std::vector<double> Wih0b((size_t)3 * g * 4, 0.0);
  for (int r = 0; r < 3 * g; ++r) {
    for (int j = 0; j < nin; ++j) Wih0b[(size_t)r * 4 + j] = Wih0[(size_t)r * nin + j];
    Wih0b[(size_t)r * 4 + 3] = bih0[r] + (r < 2 * g ? bhh0[r] : 0.0);
  }
  const auto Wih0p = pack_gru_chunked(Wih0b.data(), 4, 4, g), Whh0p = pack_gru_chunked(Whh0, g, g, g);
  const auto Wih1p = pack_gru_chunked(Wih1, g, g, g), Whh1p = pack_gru_chunked(Whh1, g, g, g);
  const auto Wop = pack_A(Wo, g, g, identity_rows(2));
  std::vector<double> brz1(2 * g);
  for (int r = 0; r < 2 * g; ++r) brz1[r] = bih1[r] + bhh1[r];
  std::vector<Acc> h0(GT), h1(GT), hn(GT);
  for (int j = 0; j < GT; ++j) {
    std::memset(h0[j], 0, sizeof(Acc));
    std::memset(h1[j], 0, sizeof(Acc));
  }
  auto sig = [](double x) { return 1.0 / (1.0 + std::exp(-x)); };
  auto chunk = [&](const std::vector<double>& W, int j, std::vector<Acc>& hsrc, Acc c0, Acc c1, Acc c2) {
    double b[64];
    for (int ks = 0; ks < KS; ++ks) {
      for (int l = 0; l < 64; ++l) b[l] = hsrc[ks >> 2][l][ks & 3];
      mfma(&W[(((size_t)j * KS + ks) * 3 + 0) * 64], b, c0);
      mfma(&W[(((size_t)j * KS + ks) * 3 + 1) * 64], b, c1);
      mfma(&W[(((size_t)j * KS + ks) * 3 + 2) * 64], b, c2);
    }
  };
  auto gates = [&](Acc ar, Acc az, Acc ain, Acc ahn, Acc hold, Acc hnew) {
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const double rg = sig(ar[l][r]), zg = sig(az[l][r]);
        const double ng = std::tanh(ain[l][r] + rg * ahn[l][r]);
        hnew[l][r] = (1.0 - zg) * ng + zg * hold[l][r];
      }
  };
  for (int s = 0; s < B; ++s) {
    double xin[64];
    for (int l = 0; l < 64; ++l) {
      const int q = l >> 4, c = l & 15;
      xin[l] = q < nin ? win[((size_t)c * B + (B - 1 - s)) * nin + q] : (q == 3 ? 1.0 : 0.0);
    }
    for (int j = 0; j < GT; ++j) {
      Acc ar, az, ain, ahn;
      std::memset(ar, 0, sizeof(Acc));
      std::memset(az, 0, sizeof(Acc));
      std::memset(ain, 0, sizeof(Acc));
      mfma(&Wih0p[((size_t)j * 3 + 0) * 64], xin, ar);
      mfma(&Wih0p[((size_t)j * 3 + 1) * 64], xin, az);
      mfma(&Wih0p[((size_t)j * 3 + 2) * 64], xin, ain);
      bias_tile(bhh0 + 2 * g, j, ahn);
      if (s > 0) chunk(Whh0p, j, h0, ar, az, ahn);
      gates(ar, az, ain, ahn, h0[j], hn[j]);
    }
    for (int j = 0; j < GT; ++j) std::memcpy(h0[j], hn[j], sizeof(Acc));
    for (int j = 0; j < GT; ++j) {
      Acc ar, az, ain, ahn;
      bias_tile(brz1.data(), j, ar);
      bias_tile(brz1.data(), GT + j, az);
      bias_tile(bih1 + 2 * g, j, ain);
      bias_tile(bhh1 + 2 * g, j, ahn);
      chunk(Wih1p, j, h0, ar, az, ain);
      if (s > 0) chunk(Whh1p, j, h1, ar, az, ahn);
      gates(ar, az, ain, ahn, h1[j], hn[j]);
    }
    for (int j = 0; j < GT; ++j) std::memcpy(h1[j], hn[j], sizeof(Acc));
  }
  Acc o;
  std::memset(o, 0, sizeof(Acc));
  double b[64];
  for (int ks = 0; ks < KS; ++ks) {
    for (int l = 0; l < 64; ++l) b[l] = h1[ks >> 2][l][ks & 3];
    mfma(&Wop[(size_t)ks * 64], b, o);
  }
  for (int l = 0; l < 64; ++l)
    if ((l >> 4) < 2) out[(l & 15) * 2 + (l >> 4)] = o[l][0] + bo[l >> 4];
}

// The same encoder with its hidden-state GEMMs as the int8-sliced kernel evaluates them (csrc/kernels_gru_i8.hip): weights from
// the 36-block stream of pack_gru_i8_stream in the kernel's consumption order, states cut into seven signed digits of
// rint(h 2^54), one integer dot product per digit pair (v_mfma_i32_16x16x64_i8: byte b of lane group kq on both operands is the
// same K entry; accumulator register r of lane group q is tile row 4 q + r), level sums recombined as the kernel does.
static void i8_tile(const signed char* blk, const std::vector<Acc>& h, long long lev[64][4][13]) {
  // B operand: lane (kq, n), byte b <-> entry of chunk b >> 2, register b & 3
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int q = l >> 4, n = l & 15, m = 4 * q + r;  // tile row of this accumulator entry
      for (int kq = 0; kq < 4; ++kq)
        for (int b = 0; b < 16; ++b) {
          signed char dh[kI8Digits];
          i8_digits(h[b >> 2][16 * kq + n][b & 3], dh);
          for (int i = 0; i < kI8Digits; ++i) {
            const signed char dw = blk[(((size_t)i * 64) + (16 * kq + m)) * 16 + b];
            for (int j = 0; j < kI8Digits; ++j)
              if (i + j >= kI8Lmin) lev[l][r][i + j] += (long long)dw * dh[j];
          }
        }
    }
}
static void i8_recombine(long long lev[64][4][13], const signed char* blk, bool merged, bool with_bias, Acc pre) {
  const double* tail = (const double*)(blk + (size_t)kI8Digits * 64 * 16);
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int f = 4 * r + (l >> 4);
      double s = 0.0;
      if (merged) {
        bool first = true;
        for (int L = kI8Lmin; L <= 12; L += 2) {
          const double m = (double)(L + 1 <= 12 ? lev[l][r][L + 1] * 256 + lev[l][r][L] : lev[l][r][L]);
          s = first ? m : std::fma(s, 0x1p-16, m);
          first = false;
        }
      } else {
        for (int L = kI8Lmin; L <= 12; ++L) s = (L == kI8Lmin) ? (double)lev[l][r][L] : std::fma(s, 0x1p-8, (double)lev[l][r][L]);
      }
      pre[l][r] = std::fma(s, tail[f], with_bias ? tail[16 + f] : pre[l][r]);
    }
}
void emu_gru_i8(const double* Wih0, const double* Whh0, const double* bih0, const double* bhh0, const double* Wih1,
                const double* Whh1, const double* bih1, const double* bhh1, const double* Wo, const double* bo, int g,
                int nin, int B, const double* win, double* out) {
  const int GT = g / 16, KS = g / 4;
  std::vector<double> Wih0b((size_t)3 * g * 4, 0.0);
  for (int r = 0; r < 3 * g; ++r) {
    for (int j = 0; j < nin; ++j) Wih0b[(size_t)r * 4 + j] = Wih0[(size_t)r * nin + j];
    Wih0b[(size_t)r * 4 + 3] = bih0[r] + (r < 2 * g ? bhh0[r] : 0.0);
  }
  const auto Wih0p = pack_gru_chunked(Wih0b.data(), 4, 4, g);
  const auto Wop = pack_A(Wo, g, g, identity_rows(2));
  std::vector<double> brz1(2 * g);
  for (int r = 0; r < 2 * g; ++r) brz1[r] = bih1[r] + bhh1[r];
  const std::vector<signed char> st = pack_gru_i8_stream(Whh0, Wih1, Whh1, bhh0 + 2 * g, brz1.data(), bih1 + 2 * g, bhh1 + 2 * g, g);
  auto block = [&](int t) { return st.data() + (size_t)t * kI8BlockBytes; };
  std::vector<Acc> h0(GT), h1(GT), hn(GT);
  for (int j = 0; j < GT; ++j) {
    std::memset(h0[j], 0, sizeof(Acc));
    std::memset(h1[j], 0, sizeof(Acc));
  }
  auto sig = [](double x) { return 1.0 / (1.0 + std::exp(-x)); };
  auto gates = [&](Acc ar, Acc az, Acc ain, Acc ahn, Acc hold, Acc hnew) {
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const double rg = sig(ar[l][r]), zg = sig(az[l][r]);
        const double ng = std::tanh(ain[l][r] + rg * ahn[l][r]);
        hnew[l][r] = (1.0 - zg) * ng + zg * hold[l][r];
      }
  };
  static long long lev[64][4][13];
  for (int s = 0; s < B; ++s) {
    double xin[64];
    for (int l = 0; l < 64; ++l) {
      const int q = l >> 4, c = l & 15;
      xin[l] = q < nin ? win[((size_t)c * B + (B - 1 - s)) * nin + q] : (q == 3 ? 1.0 : 0.0);
    }
    for (int j = 0; j < GT; ++j) {
      Acc a3[3], ahn;
      for (int gk = 0; gk < 3; ++gk) {
        std::memset(a3[gk], 0, sizeof(Acc));
        mfma(&Wih0p[((size_t)j * 3 + gk) * 64], xin, a3[gk]);
      }
      bias_tile(bhh0 + 2 * g, j, ahn);
      if (s > 0)
        for (int gk = 0; gk < 3; ++gk) {
          std::memset(lev, 0, sizeof(lev));
          i8_tile(block(3 * j + gk), h0, lev);
          i8_recombine(lev, block(3 * j + gk), true, gk == 2, gk == 2 ? ahn : a3[gk]);
        }
      gates(a3[0], a3[1], a3[2], ahn, h0[j], hn[j]);
    }
    for (int j = 0; j < GT; ++j) std::memcpy(h0[j], hn[j], sizeof(Acc));
    for (int j = 0; j < GT; ++j) {
      Acc pre[4];  // r, z, n input side, n hidden side
      bias_tile(bhh1 + 2 * g, j, pre[3]);
      for (int gk = 0; gk < 3; ++gk) {
        const int t = 12 + 6 * j + 2 * gk;  // hidden-side block, then input-side block
        std::memset(lev, 0, sizeof(lev));
        if (s > 0) i8_tile(block(t), h1, lev);
        if (gk == 2) {
          if (s > 0) i8_recombine(lev, block(t), true, true, pre[3]);
          std::memset(lev, 0, sizeof(lev));
        }
        i8_tile(block(t + 1), h0, lev);
        i8_recombine(lev, block(t + 1), gk == 2, true, pre[gk]);
      }
      gates(pre[0], pre[1], pre[2], pre[3], h1[j], hn[j]);
    }
    for (int j = 0; j < GT; ++j) std::memcpy(h1[j], hn[j], sizeof(Acc));
  }
  Acc o;
  std::memset(o, 0, sizeof(Acc));
  double b[64];
  for (int ks = 0; ks < KS; ++ks) {
    for (int l = 0; l < 64; ++l) b[l] = h1[ks >> 2][l][ks & 3];
    mfma(&Wop[(size_t)ks * 64], b, o);
  }
  for (int l = 0; l < 64; ++l)
    if ((l >> 4) < 2) out[(l & 15) * 2 + (l >> 4)] = o[l][0] + bo[l >> 4];
}

// digits of rint(x 2^54) put back together (tests: exactness of the cut, the row exponent of a matrix row)
void emu_i8_roundtrip(const double* x, int n, double* back, int* top_digit) {
  for (int k = 0; k < n; ++k) {
    signed char d[kI8Digits];
    i8_digits(x[k], d);
    long long X = 0;
    for (int i = kI8Digits - 1; i >= 0; --i) X = X * 256 + d[i];
    back[k] = std::ldexp((double)X, -kI8Frac);
    top_digit[k] = d[kI8Digits - 1];
  }
}
void emu_i8_row_exponents(const double* W, int rows, int K, int* e) {
  const auto v = i8_row_exponents(W, rows, K);
  for (int r = 0; r < rows; ++r) e[r] = v[r];
}
}
