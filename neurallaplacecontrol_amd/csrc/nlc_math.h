// Bounded-range FP64 elementary functions for the NLC kernels.
//
// The hot kernels evaluate ~2000 FP64 transcendentals per (sample, horizon step).  ocml's full-range
// versions (Payne-Hanek reduction, denormal paths) are several hundred instructions each once inlined and
// blow the unroll budget of the fused kernels, so the kernels use these instead: every argument here has
// a known bounded range (tanh outputs scaled by pi, gate pre-activations, ...).  Accuracy target is a few
// ulp; tests/test_math_host.py checks them against libm on dense grids (this header also compiles as
// plain C++ for that purpose).
#pragma once
#include <math.h>
#include <stdint.h>


#if defined(__HIPCC__)
#define NLC_HD __host__ __device__ __forceinline__
#else
#define NLC_HD inline
#endif

namespace nlc {
namespace m {

NLC_HD double rcp_refined(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  // v_rcp_f64 is good to ~2^-24; one cubically convergent step r (1 + e + e^2), e = 1 - d r, leaves e^3 ~ 2^-72:
  // three FMAs instead of the four of two Newton steps
  const double r = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, r, 1.0);
  return fma(r, fma(e, e, e), r);
#else
  return 1.0 / d;
#endif
}

// n / d for finite, normal d well inside the exponent range (no scaling / fixup paths)
NLC_HD double div_fast(double n, double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double r = rcp_refined(d);
  const double q = n * r;
  return fma(fma(-d, q, n), r, q);
#else
  return n / d;
#endif
}

// expm1(r) = r + r^2 q(r) for |r| <= ln2/2.  q: degree-9 Chebyshev-interpolation polynomial of
// (e^r - 1 - r)/r^2 on the interval (near-minimax; max relative error of the expm1 result 4.1e-17 in exact
// arithmetic, vs degree 11 for the same accuracy with Taylor coefficients) -- two FMAs fewer per call.
NLC_HD double expm1_poly(double r) {
  double q = 0x1.af38a9b0ec855p-26;
  q = fma(q, r, 0x1.289185613a3d6p-22);
  q = fma(q, r, 0x1.71de0dae63bb3p-19);
  q = fma(q, r, 0x1.a019b90d2ae7ap-16);
  q = fma(q, r, 0x1.a01a01a7c41d5p-13);
  q = fma(q, r, 0x1.6c16c1788bd90p-10);
  q = fma(q, r, 0x1.11111111109b3p-7);
  q = fma(q, r, 0x1.5555555553d63p-5);
  q = fma(q, r, 0x1.5555555555556p-3);
  q = fma(q, r, 0x1.0000000000001p-1);
  return fma(q * r, r, r);
}

// y = n ln2 + r
NLC_HD double exp_reduce(double y, int* n) {
  const double kLog2e = 1.44269504088896338700e+00;
  const double kLn2Hi = 6.93147180369123816490e-01;
  const double kLn2Lo = 1.90821492927058770002e-10;
  // round-to-nearest by the 1.5 * 2^52 shift: the integer lands in the low word of the shifted sum, so no
  // v_rndne / v_cvt_i32 is needed (|y log2(e)| < 2^31 for every clamped argument)
  const double kShift = 6755399441055744.0;
  const double sh = fma(y, kLog2e, kShift);
  const double fn = sh - kShift;
  double r = fma(-fn, kLn2Hi, y);
  r = fma(-fn, kLn2Lo, r);
  int64_t bits;
  __builtin_memcpy(&bits, &sh, sizeof(bits));
  *n = (int)(uint32_t)(bits & 0xffffffffLL);
  return r;
}

// exp(y) for y in [-745, 709]
NLC_HD double exp_d(double y) {
  y = fmin(fmax(y, -745.0), 709.0);
  int n;
  const double r = exp_reduce(y, &n);
  return ldexp(1.0 + expm1_poly(r), n);
}

// expm1(y) for y <= 0
NLC_HD double expm1_neg(double y) {
  y = fmax(y, -745.0);
  int n;
  const double r = exp_reduce(y, &n);
  const double p = expm1_poly(r);
  // 2^n (1 + p) - 1 = 2^n p + (2^n - 1); exact-ish for n = 0, no cancellation for n <= -1
  const double two_n = ldexp(1.0, n);
  return fma(two_n, p, two_n - 1.0);
}

// exp(y) for y <= 0 (one-sided clamp: callers pass -|x|)
NLC_HD double exp_neg(double y) {
  y = fmax(y, -745.0);
  int n;
  const double r = exp_reduce(y, &n);
  return ldexp(1.0 + expm1_poly(r), n);
}

// (the one-line form rcp(1 + exp(-x)) is 3 instructions shorter but tips the GRU kernel, which sits at exactly
// 256 VGPRs for 2 waves/SIMD, into 308 B/lane of scratch: 3.29 -> 3.94 ms.  Measured; keep the two-sided form.)
NLC_HD double sigmoid_d(double x) {
  const double e = exp_neg(-fabs(x));  // (0, 1]
  const double inv = rcp_refined(1.0 + e);
  return x >= 0.0 ? inv : e * inv;
}

// tanh(x) = -em/(2 + em), em = e^{-2|x|} - 1.  No saturation branch: for large |x| em rounds to -1 and the
// quotient to exactly 1.  The denominator is in [1, 2], so the refined reciprocal (< 1 ulp) replaces a division.
NLC_HD double tanh_d(double x) {
  const double em = expm1_neg(-2.0 * fabs(x));  // (-1, 0]
  const double t = -em * rcp_refined(2.0 + em);
  return copysign(t, x);
}

// Two tanh values sharing ONE reciprocal: -em_a (2 + em_b) R and -em_b (2 + em_a) R with R = 1 / ((2+em_a)(2+em_b));
// both denominators are in [1, 2], so the product cannot overflow.  Three multiplies replace the second
// v_rcp_f64 + refinement.  A saturated input gives 1 within an ulp, not exactly 1 as tanh_d does, so this is for the
// hidden-layer activations only (the sphere map needs the exact saturation).
NLC_HD void tanh_pair_d(double xa, double xb, double* ta, double* tb) {
  const double ea = expm1_neg(-2.0 * fabs(xa)), eb = expm1_neg(-2.0 * fabs(xb));
  const double da = 2.0 + ea, db = 2.0 + eb;
  const double R = rcp_refined(da * db);
  *ta = copysign((-ea * db) * R, xa);
  *tb = copysign((-eb * da) * R, xb);
}

// ---- hidden-layer activations of the rollout kernels (round 3): instruction count, not ulps.
// e^{-2|x|}: one-constant reduction (|n| ln2 2^-53 <= 3e-14 where it matters), e^r = (q r + 1) r + 1 with q of degree 7
// (Chebyshev interpolation of (e^r - 1 - r)/r^2 on |r| <= ln2/2: relative error 7.4e-14), nine FMAs in one Horner chain;
// the clamp is a bare v_max_f64 on the device (fmax() first canonicalises its operand with a second v_max_f64).
NLC_HD double exp_m2abs_fast(double x) {
  double y = -2.0 * fabs(x);
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_max_f64 %0, %1, %2" : "=v"(y) : "v"(y), "s"(-745.0));
#else
  y = y < -745.0 ? -745.0 : y;
#endif
  const double kShift = 6755399441055744.0;
  const double sh = fma(y, 1.44269504088896338700e+00, kShift);
  const double fn = sh - kShift;
  const double r = fma(-fn, 6.93147180559945286227e-01, y);
  int64_t bits;
  __builtin_memcpy(&bits, &sh, sizeof(bits));
  const int n = (int)(uint32_t)(bits & 0xffffffffLL);
  double q = 0x1.72ad458027fbcp-19;
  q = fma(q, r, 0x1.a136bf03ec612p-16);
  q = fma(q, r, 0x1.a019c36bc053cp-13);
  q = fma(q, r, 0x1.6c166bde96885p-10);
  q = fma(q, r, 0x1.111111170bc08p-7);
  q = fma(q, r, 0x1.55555565c7e0ep-5);
  q = fma(q, r, 0x1.5555555554f96p-3);
  q = fma(q, r, 0x1.fffffffffe062p-2);
  q = fma(q, r, 1.0);
  q = fma(q, r, 1.0);
  return ldexp(q, n);
}
// Two tanh values, t = (1 - e)/(1 + e) with e = e^{-2|x|}, ONE reciprocal for both (denominators in [1, 2]).  Absolute
// error <= 1.5e-13 (the parity bar is 1e-5, the tests hold 1e-9); a saturated input gives 1 within an ulp.  Hidden
// layers only: the sphere map keeps tanh_d (exact saturation, few ulp).
NLC_HD void tanh_pair_fast(double xa, double xb, double* ta, double* tb) {
  const double ea = exp_m2abs_fast(xa), eb = exp_m2abs_fast(xb);
  const double da = 1.0 + ea, db = 1.0 + eb;
  const double R = rcp_refined(da * db);
  *ta = copysign(((1.0 - ea) * db) * R, xa);
  *tb = copysign(((1.0 - eb) * da) * R, xb);
}

// fdlibm __kernel_sin / __kernel_cos on |y| <= pi/4 (+ a few ulp)
NLC_HD double sin_poly(double y) {
  const double z = y * y;
  double p = 1.58969099521155010221e-10;
  p = fma(p, z, -2.50507602534068634195e-08);
  p = fma(p, z, 2.75573137070700676789e-06);
  p = fma(p, z, -1.98412698298579493134e-04);
  p = fma(p, z, 8.33333333332248946124e-03);
  p = fma(p, z, -1.66666666666666324348e-01);
  return fma(y * z, p, y);
}
NLC_HD double cos_poly(double y) {
  const double z = y * y;
  double p = -1.13596475577881948265e-11;
  p = fma(p, z, 2.08757232129817482790e-09);
  p = fma(p, z, -2.75573143513906633035e-07);
  p = fma(p, z, 2.48015872894767294178e-05);
  p = fma(p, z, -1.38888888888741095749e-03);
  p = fma(p, z, 4.16666666666666019037e-02);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  // 1 - z/2 + z^2 p with the rounding error of (1 - hz) folded back in (as fdlibm does)
  return w + fma(z * z, p, (1.0 - w) - hz);  // (explicit: the same bits under -ffp-contract=off and =fast)
}

constexpr double kPio2Hi = 1.57079632679489655800e+00;
constexpr double kPio2Lo = 6.12323399573676603587e-17;

// sin and cos of x, |x| <= ~2 pi (quadrant count small enough for a two-constant Cody-Waite step)
NLC_HD void sincos_bounded(double x, double* s, double* c) {
  const double fk = rint(x * 6.36619772367581382433e-01);  // 2/pi
  double y = fma(-fk, kPio2Hi, x);
  y = fma(-fk, kPio2Lo, y);
  const int k = (int)fk;
  const double sy = sin_poly(y), cy = cos_poly(y);
  const double ss = (k & 1) ? cy : sy;
  const double cc = (k & 1) ? sy : cy;
  *s = (k & 2) ? -ss : ss;
  *c = ((k + 1) & 2) ? -cc : cc;
}

// tan(x) for x in [0, pi/2]
NLC_HD double tan_0_halfpi(double x) {
  const bool hi = x > 0.78539816339744830962;
  // b = pi/2 - x computed with the two-constant split
  const double b = (kPio2Hi - x) + kPio2Lo;
  const double y = hi ? b : x;
  const double sy = sin_poly(y), cy = cos_poly(y);
  return hi ? div_fast(cy, sy) : div_fast(sy, cy);
}

// tan(x) for x in [0, pi/2] as (cos a + sin a)/(cos a - sin a) with a = x - pi/4 in [-pi/4, pi/4]: the
// polynomials' range, so no branch.  x is the reference's rounded phi/2 + pi/4 and a = (x - pi4_hi) - pi4_lo
// recovers it to ~1e-33, so the result tracks tan(x) of that very double (the dominant error of the sphere map is
// the rounding of x itself).  num/den are returned separately so a caller can fold the division into a later
// product.  den is clamped at the value that reproduces tan of the double nearest pi/2 (1.633e16), the
// reference's saturated |F| when tanh rounds to 1.
NLC_HD void tan_parts_0_halfpi(double x, double* num, double* den) {
  const double a = (x - 0.5 * kPio2Hi) - 0.5 * kPio2Lo;
  const double sa = sin_poly(a), ca = cos_poly(a);
  *num = ca + sa;
  *den = fmax(ca - sa, 8.659560562354934e-17);
}
NLC_HD double tan_pi4_plus(double x) {
  double num, den;
  tan_parts_0_halfpi(x, &num, &den);
  return div_fast(num, den);
}

// cos(x + j0*pi/2) for |x| <= ~2 pi and an integer quadrant offset j0 (exact phase i^k of the Fourier ILT)
NLC_HD double cos_quadrant(double x, int j0) {
  const double fk = rint(x * 6.36619772367581382433e-01);
  double y = fma(-fk, kPio2Hi, x);
  y = fma(-fk, kPio2Lo, y);
  const int j = (int)fk + j0;
  const double sy = sin_poly(y), cy = cos_poly(y);
  const double cc = (j & 1) ? sy : cy;
  return ((j + 1) & 2) ? -cc : cc;
}

// ---- short forms for the stand-alone Fourier ILT kernel, which is bound by its FP64 instruction count
// (kernels_ilt.hip).  Chebyshev-interpolation polynomials in z = y^2; ABSOLUTE accuracy ~2e-16 (the fdlibm
// kernels above are relative-accurate near the zeros and cost 21 instructions for the sin/cos pair).
// Every constant comes from an IltTrigK the caller holds: the kernel pins its members in SGPRs, so each Horner step
// is ONE v_fma_f64 with a scalar addend (left to itself the compiler parks the 64-bit constants in VGPRs and emits a
// v_mov_b64 + v_fmac_f64 pair per step, and at 128 VGPRs it spills).
struct IltTrigK {
  double s[6];   // sin(a)/a - 1 = z S(z), |a| <= pi/4: fdlibm S1..S6, highest degree first
  double c4[6];  // cos(a) = 1 + z C(z), |a| <= pi/4 (+0.1 %): degree 6 in z, approximation error 4.8e-17
  double c2[8];  // cos(y) = 1 + z C(z), |y| <= pi/2 (+0.01 %): degree 8 in z, approximation error 3.9e-18
  double pio4_hi, pio4_lo, pio2_hi, pio2_lo, inv_pi, round_shift, den_min;
};
constexpr double kRoundShift = 6755399441055744.0;  // 1.5 * 2^52
NLC_HD IltTrigK ilt_trig_k() {
  IltTrigK K = {
      {1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06,
       -1.98412698298579493134e-04, 8.33333333332248946124e-03, -1.66666666666666324348e-01},
      {0x1.1b893bd61e9c0p-29, -0x1.27df3a2ecce24p-22, 0x1.a019f7eebbb5ep-16, -0x1.6c16c163b498ep-10,
       0x1.555555554e6d2p-5, -0x1.fffffffffff77p-2},
      {0x1.9f230043b0248p-45, -0x1.93509d0fccb4ap-37, 0x1.1eecdf1eb9cc7p-29, -0x1.27e4f978cae28p-22,
       0x1.a01a01994a786p-16, -0x1.6c16c16c09b0ep-10, 0x1.55555555553c4p-5, -0x1.ffffffffffffbp-2},
      0.5 * kPio2Hi, 0.5 * kPio2Lo, kPio2Hi, kPio2Lo, 3.18309886183790691216e-01, kRoundShift,
      // tan of the double nearest pi/2 (1.633e16) = the reference's saturated |F| when tanh rounds to 1
      8.659560562354934e-17};
  return K;
}
// cos(x + m pi/2) for m in {0, 1} and |x| < ~1e5: ONE reduction by pi, q = rint(x/pi + m/2), y = x - (2q - m) pi/2 in
// [-pi/2, pi/2], result (-1)^q cos(y).  The rounding uses the 1.5 * 2^52 shift, so the parity of q is bit 0 of the
// shifted sum's low word and flips the result's sign bit directly (no v_rndne / v_cvt / compare / select).
// half_m = m/2 and dm = (double)m are lane constants of the caller.  (m/2 cannot ride in the shift constant: at
// 1.5 * 2^52 the spacing is 1.)
NLC_HD double cos_plus_mpio2(const IltTrigK& K, double x, double half_m, double dm) {
  const double sh = fma(x, K.inv_pi, half_m) + K.round_shift;
  const double q = sh - K.round_shift;
  const double n = fma(2.0, q, -dm);
  double y = fma(-n, K.pio2_hi, x);
  y = fma(-n, K.pio2_lo, y);
  const double z = y * y;
  double p = fma(K.c2[0], z, K.c2[1]);
  for (int i = 2; i < 8; ++i) p = fma(p, z, K.c2[i]);
  const double c = fma(p, z, 1.0);
  uint64_t sb, cb;
  __builtin_memcpy(&sb, &sh, sizeof(sb));
  __builtin_memcpy(&cb, &c, sizeof(cb));
  cb ^= sb << 63;
  double r;
  __builtin_memcpy(&r, &cb, sizeof(r));
  return r;
}
// sin(x + m pi/2) and cos(x + m pi/2) by the same single reduction (backward of the Fourier ILT needs both):
// sin(y) = y + y z S(z) on |y| <= pi/2, degree 7 in z (approximation error 1.3e-18)
NLC_HD void sincos_plus_mpio2(const IltTrigK& K, double x, double half_m, double dm, double* sn, double* cs) {
  const double sh = fma(x, K.inv_pi, half_m) + K.round_shift;
  const double q = sh - K.round_shift;
  const double n = fma(2.0, q, -dm);
  double y = fma(-n, K.pio2_hi, x);
  y = fma(-n, K.pio2_lo, y);
  const double z = y * y;
  double p = fma(K.c2[0], z, K.c2[1]);
  for (int i = 2; i < 8; ++i) p = fma(p, z, K.c2[i]);
  const double c = fma(p, z, 1.0);
  double u = 0x1.89a3f16388edcp-49;
  u = fma(u, z, -0x1.ae513415aadccp-41);
  u = fma(u, z, 0x1.6124014cbe2fcp-33);
  u = fma(u, z, -0x1.ae6455a1a9e7ep-26);
  u = fma(u, z, 0x1.71de3a54562a8p-19);
  u = fma(u, z, -0x1.a01a01a018aa6p-13);
  u = fma(u, z, 0x1.1111111111107p-7);
  u = fma(u, z, -0x1.5555555555555p-3);
  const double sv = fma(y * z, u, y);
  uint64_t sb, cb, vb;
  __builtin_memcpy(&sb, &sh, sizeof(sb));
  __builtin_memcpy(&cb, &c, sizeof(cb));
  __builtin_memcpy(&vb, &sv, sizeof(vb));
  cb ^= sb << 63;
  vb ^= sb << 63;
  __builtin_memcpy(cs, &cb, sizeof(cb));
  __builtin_memcpy(sn, &vb, sizeof(vb));
}
// tan(x) = num/den for x in [0, pi/2] as (cos a + sin a)/(cos a - sin a), a = x - pi/4 (cf. tan_parts_0_halfpi);
// den is clamped at the value that reproduces the reference's saturated |F|.
NLC_HD void tan_parts_short(const IltTrigK& K, double x, double* num, double* den) {
  const double a = (x - K.pio4_hi) - K.pio4_lo;
  const double z = a * a;
  double p = fma(K.s[0], z, K.s[1]);
  for (int i = 2; i < 6; ++i) p = fma(p, z, K.s[i]);
  const double sa = fma(a * z, p, a);
  double c = fma(K.c4[0], z, K.c4[1]);
  for (int i = 2; i < 6; ++i) c = fma(c, z, K.c4[i]);
  const double ca = fma(c, z, 1.0);
  *num = ca + sa;
  *den = fmax(ca - sa, K.den_min);
}

// ---- round 6: forms of the row-per-lane Fourier ILT kernel (kernels_ilt.hip: ilt_fourier_rows_kernel), where the term index --
// hence the quarter turn of its phase -- is a compile-time constant and the instruction count is what bounds the stream.
//   tan(pi/4 + a) = (1 + tan a)/(1 - tan a) with tan a = a (Q + z P)/Q, z = a^2, |a| <= pi/4: the rational form of Cephes' tan.c
//   (P of degree 2, monic Q of degree 4; relative error 3e-16), written for num/den directly and with both polynomials negated
//   so that num, den > 0:  num = Qn + a (Qn + z Pn), den = Qn - a (Qn + z Pn).  Twelve instructions against the eighteen of the
//   sin / cos pair (tan_parts_short).
//   cos(x + m pi/2), m known: ONE reduction by pi (q = rint(x / pi) rides in a single FMA with the 1.5 2^52 shift), then the cosine
//   (m = 0) or the sine (m = 1: cos(x + pi/2) = -sin x) polynomial on [-pi/2, pi/2]; the caller folds the constant sign.
struct IltRowK {
  double qn[4];   // Qn(z) = -(z^4 + Q0 z^3 + Q1 z^2 + Q2 z + Q3): the four lower coefficients, negated
  double pn[3];   // Pn(z) = -(P0 z^2 + P1 z + P2)
  double c2[8];   // cos(y) = 1 + z C(z), |y| <= pi/2 (IltTrigK::c2)
  double s2[8];   // sin(y) = y + y z S(z), |y| <= pi/2, degree 7 in z (sincos_plus_mpio2)
  double pio4_hi, pio4_lo, pi_hi, pi_lo, inv_pi, round_shift, den_min;
};
NLC_HD IltRowK ilt_row_k() {
  const IltTrigK T = ilt_trig_k();
  IltRowK K = {
      {-1.36812963470692954678e4, 1.32089234440210967447e6, -2.50083801823357915839e7, 5.38695755929454629881e7},
      {1.30936939181383777646e4, -1.15351664838587416140e6, 1.79565251976484877988e7},
      {T.c2[0], T.c2[1], T.c2[2], T.c2[3], T.c2[4], T.c2[5], T.c2[6], T.c2[7]},
      {0x1.89a3f16388edcp-49, -0x1.ae513415aadccp-41, 0x1.6124014cbe2fcp-33, -0x1.ae6455a1a9e7ep-26, 0x1.71de3a54562a8p-19,
       -0x1.a01a01a018aa6p-13, 0x1.1111111111107p-7, -0x1.5555555555555p-3},
      T.pio4_hi, T.pio4_lo, 2.0 * kPio2Hi, 2.0 * kPio2Lo, T.inv_pi, T.round_shift,
      // num / den at a = pi/4 must be the reference's saturated |F| = tan(double nearest pi/2) = 1.633e16 (IltTrigK::den_min):
      // num there is 2 Qn((pi/4)^2) = 7.7885e7
      7.788508645242503e7 / 1.633123935319537e16};
  return K;
}
NLC_HD void tan_parts_rat(const IltRowK& K, double a, double* num, double* den) {
  const double z = a * a;
  double q = K.qn[0] - z;
  q = fma(q, z, K.qn[1]);
  q = fma(q, z, K.qn[2]);
  q = fma(q, z, K.qn[3]);
  double p = fma(K.pn[0], z, K.pn[1]);
  p = fma(p, z, K.pn[2]);
  const double xt = a * fma(z, p, q);
  *num = q + xt;
  *den = fmax(q - xt, K.den_min);
}
// (-1)^q cos(y) for ODD = 0, (-1)^q sin(y) for ODD = 1, where x = q pi + y: the caller's term is +this (ODD = 0) or -this (ODD = 1)
template <int ODD>
NLC_HD double cos_or_sin_reduced(const IltRowK& K, double x) {
  const double sh = fma(x, K.inv_pi, K.round_shift);
  const double q = sh - K.round_shift;
  double y = fma(-q, K.pi_hi, x);
  y = fma(-q, K.pi_lo, y);
  const double z = y * y;
  double r;
  if (ODD) {
    double u = fma(K.s2[0], z, K.s2[1]);
    for (int i = 2; i < 8; ++i) u = fma(u, z, K.s2[i]);
    r = fma(y * z, u, y);
  } else {
    double c = fma(K.c2[0], z, K.c2[1]);
    for (int i = 2; i < 8; ++i) c = fma(c, z, K.c2[i]);
    r = fma(c, z, 1.0);
  }
  uint64_t sb, rb;
  __builtin_memcpy(&sb, &sh, sizeof(sb));
  __builtin_memcpy(&rb, &r, sizeof(rb));
  rb ^= sb << 63;
  __builtin_memcpy(&r, &rb, sizeof(r));
  return r;
}

// (-1)^q sin(y) and (-1)^q cos(y), x = q pi + y: sin x and cos x by the one reduction of cos_or_sin_reduced (the backward of
// the row-per-lane Fourier ILT kernel needs both of every term)
NLC_HD void sincos_reduced(const IltRowK& K, double x, double* sn, double* cs) {
  const double sh = fma(x, K.inv_pi, K.round_shift);
  const double q = sh - K.round_shift;
  double y = fma(-q, K.pi_hi, x);
  y = fma(-q, K.pi_lo, y);
  const double z = y * y;
  double u = fma(K.s2[0], z, K.s2[1]);
  for (int i = 2; i < 8; ++i) u = fma(u, z, K.s2[i]);
  double sv = fma(y * z, u, y);
  double c = fma(K.c2[0], z, K.c2[1]);
  for (int i = 2; i < 8; ++i) c = fma(c, z, K.c2[i]);
  double cv = fma(c, z, 1.0);
  uint64_t sb, vb, cb;
  __builtin_memcpy(&sb, &sh, sizeof(sb));
  __builtin_memcpy(&vb, &sv, sizeof(vb));
  __builtin_memcpy(&cb, &cv, sizeof(cb));
  vb ^= sb << 63;
  cb ^= sb << 63;
  __builtin_memcpy(sn, &vb, sizeof(vb));
  __builtin_memcpy(cs, &cb, sizeof(cb));
}

}  // namespace m
}  // namespace nlc
