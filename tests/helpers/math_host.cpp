// Host build of neurallaplacecontrol_amd/csrc/nlc_math.h for tests/test_math_host.py (g++, no GPU).
#include "../../neurallaplacecontrol_amd/csrc/nlc_math.h"
#include "../../tools/nlc_math_table.h"  // the measured-and-dropped table-driven alternative (not in the product)
extern "C" {
void nlc_t_tanh(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = nlc::m::tanh_d(x[i]); }
void nlc_t_sigmoid(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = nlc::m::sigmoid_d(x[i]); }
void nlc_t_exp(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = nlc::m::exp_d(x[i]); }
void nlc_t_sin(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) { double s, c; nlc::m::sincos_bounded(x[i], &s, &c); y[i] = s; } }
void nlc_t_cos(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) { double s, c; nlc::m::sincos_bounded(x[i], &s, &c); y[i] = c; } }
void nlc_t_tan(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = nlc::m::tan_0_halfpi(x[i]); }
}
extern "C" {
void nlc_t_tan_pi4(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = nlc::m::tan_pi4_plus(x[i]); }
void nlc_t_cosq(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = nlc::m::cos_quadrant(x[i], (int)(i % 7) - 3); }
}
static const double kTab[64] = {NLC_EXP_TABLE_VALUES};
extern "C" {
void nlc_t_tanh_t(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = nlc::m::tanh_t(x[i], kTab); }
void nlc_t_sigmoid_t(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = nlc::m::sigmoid_t(x[i], kTab); }
}
extern "C" {
// short forms of the Fourier ILT kernel: cos(x + m pi/2) with m = i & 1, tan(x) on [0, pi/2] as num/den
void nlc_t_cos_mpio2(const double* x, double* y, long n) {
  for (long i = 0; i < n; ++i) {
    const double dm = (double)(i & 1);
    y[i] = nlc::m::cos_plus_mpio2(nlc::m::ilt_trig_k(), x[i], 0.5 * dm, dm);
  }
}
void nlc_t_tan_short(const double* x, double* y, long n) {
  for (long i = 0; i < n; ++i) {
    double num, den;
    nlc::m::tan_parts_short(nlc::m::ilt_trig_k(), x[i], &num, &den);
    y[i] = num / den;
  }
}
void nlc_t_sin_mpio2(const double* x, double* y, long n) {
  for (long i = 0; i < n; ++i) {
    const double dm = (double)(i & 1);
    double sn, cs;
    nlc::m::sincos_plus_mpio2(nlc::m::ilt_trig_k(), x[i], 0.5 * dm, dm, &sn, &cs);
    y[i] = sn;
  }
}
}
extern "C" {
// round 3: the rollout kernels' hidden-layer activation (instruction count over ulps): pairs (x[i], x[i + 1])
void nlc_t_tanh_pair_fast(const double* x, double* y, long n) {
  for (long i = 0; i + 1 < n; i += 2) nlc::m::tanh_pair_fast(x[i], x[i + 1], &y[i], &y[i + 1]);
  if (n & 1) { double t; nlc::m::tanh_pair_fast(x[n - 1], 0.0, &y[n - 1], &t); }
}
}
extern "C" {
// round 6: the row-per-lane Fourier ILT kernel's forms -- tan(pi/4 + a) as num/den (Cephes rational), cos(x + m pi/2) with m = i & 1
void nlc_t_tan_rat(const double* a, double* y, long n) {
  for (long i = 0; i < n; ++i) {
    double num, den;
    nlc::m::tan_parts_rat(nlc::m::ilt_row_k(), a[i], &num, &den);
    y[i] = num / den;
  }
}
void nlc_t_cos_kpio2(const double* x, double* y, long n) {
  for (long i = 0; i < n; ++i)
    y[i] = (i & 1) ? -nlc::m::cos_or_sin_reduced<1>(nlc::m::ilt_row_k(), x[i]) : nlc::m::cos_or_sin_reduced<0>(nlc::m::ilt_row_k(), x[i]);
}
}
extern "C" {
void nlc_t_sincos_reduced(const double* x, double* sn, double* cs, long n) {
  for (long i = 0; i < n; ++i) nlc::m::sincos_reduced(nlc::m::ilt_row_k(), x[i], &sn[i], &cs[i]);
}
}
