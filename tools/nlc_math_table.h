// Table-driven exp / tanh / sigmoid: the measured-and-dropped alternative to the polynomial forms of
// neurallaplacecontrol_amd/csrc/nlc_math.h (kept out of the product; round-1 measurement in the comment below).
#pragma once
#include "../neurallaplacecontrol_amd/csrc/nlc_math.h"
#include "nlc_exp_table.h"
namespace nlc {
namespace m {
// ---- table-driven variants: e^y = 2^n * T[j] * e^r with T[j] = 2^(j/64) (64-entry table, LDS on the device),
// |r| <= ln2/128, so a degree-5 polynomial replaces the degree-13 one: ~6 fewer FP64 instructions per call.
// MEASURED AND NOT USED by the kernels (MI355X, cfg2): GRU 3.43 -> 3.38 ms but rollout 1.41 -> 1.49 ms -- the
// ds_read in the middle of every dependent chain costs more than the shorter polynomial saves at 1 wave/SIMD.
// Kept (with its host test) as the documented alternative.
NLC_HD double exp_reduce_t(double y, const double* tab, int* n, double* T) {
  const double kf = rint(y * k64_Ln2);
  double r = fma(-kf, kLn2_64Hi, y);
  r = fma(-kf, kLn2_64Lo, r);
  const int ki = (int)kf;
  *n = ki >> 6;
  *T = tab[ki & 63];
  return r;
}
// e^r - 1 for |r| <= ln2/128 (remainder r^6/720 < 4e-17)
NLC_HD double expm1_poly5(double r) {
  double q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  q = fma(q, r, 1.0 / 6.0);
  q = fma(q, r, 0.5);
  return fma(q * r, r, r);
}
// exp(y), y <= 0 (callers pass -|x|); underflows to 0 through ldexp for y < -745
NLC_HD double exp_neg_t(double y, const double* tab) {
  y = fmax(y, -750.0);
  int n;
  double T;
  const double r = exp_reduce_t(y, tab, &n, &T);
  return ldexp(fma(T, expm1_poly5(r), T), n);
}
// e^r - 1 for |r| <= ln2/128 to a RELATIVE 5e-18 (one more term: the result itself can be as small as r)
NLC_HD double expm1_poly6(double r) {
  double q = fma(r, 1.0 / 720.0, 1.0 / 120.0);
  q = fma(q, r, 1.0 / 24.0);
  q = fma(q, r, 1.0 / 6.0);
  q = fma(q, r, 0.5);
  return fma(q * r, r, r);
}
// expm1(y), y <= 0: S (1 + p) - 1 = S p + (S - 1) with S = 2^n T[j]; S - 1 is exact for n >= -1
NLC_HD double expm1_neg_t(double y, const double* tab) {
  y = fmax(y, -750.0);
  int n;
  double T;
  const double r = exp_reduce_t(y, tab, &n, &T);
  const double S = ldexp(T, n);
  return fma(S, expm1_poly6(r), S - 1.0);
}
NLC_HD double sigmoid_t(double x, const double* tab) {
  const double e = exp_neg_t(-fabs(x), tab);  // (0, 1]
  const double inv = rcp_refined(1.0 + e);
  return x >= 0.0 ? inv : e * inv;
}
NLC_HD double tanh_t(double x, const double* tab) {
  const double a = fabs(x);
  const double em = expm1_neg_t(-2.0 * a, tab);  // e^{-2a} - 1 in (-1, 0]
  double t = div_fast(-em, 2.0 + em);
  t = a > 20.0 ? 1.0 : t;
  return copysign(t, x);
}

}  // namespace m
}  // namespace nlc
