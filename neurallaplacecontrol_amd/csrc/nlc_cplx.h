// Complex FP64 helpers of the de Hoog kernels (forward: kernels_dehoog.hip, backward: kernels_dehoog_bwd.hip).
#pragma once
#include "nlc_device.h"

namespace nlc {

struct cplx {
  double re, im;
};
__device__ __forceinline__ cplx cadd(cplx a, cplx b) {
#pragma clang fp contract(off)
  return {a.re + b.re, a.im + b.im};
}
__device__ __forceinline__ cplx csub(cplx a, cplx b) {
#pragma clang fp contract(off)
  return {a.re - b.re, a.im - b.im};
}
// Every function here pins `fp contract(off)` and spells the fused multiply-adds it wants: under -ffp-contract=fast the compiler
// picks which product of a*b - c*d to fuse from the surrounding code (use counts), and it re-fuses explicit fma() calls with
// neighbouring contractable adds (fadd (fma x y (fmul u v)) z -> fma x y (fma u v z)), so the same source inlined into two
// kernels can round differently -- the persistent step chain (kernels_dehoog_chain.hip) is tested bit for bit against the
// staged launches (round 4).
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
#pragma clang fp contract(off)
  return {fma(a.re, b.re, -(a.im * b.im)), fma(a.re, b.im, a.im * b.re)};
}
// a / b = a conj(b) / |b|^2 with ONE refined reciprocal (v_rcp_f64 + one cubic refinement step, <= 1 ulp) instead of two IEEE
// divisions (v_div_scale / v_div_fmas / v_div_fixup sequences with their VCC hazards): the QD table needs ~M^2 of
// these per row, and they were 55 % of the kernel's issue slots.
__device__ __forceinline__ cplx cdiv(cplx a, cplx b) {
#pragma clang fp contract(off)
  const double inv = m::rcp_refined(fma(b.re, b.re, b.im * b.im));
  return {fma(a.re, b.re, a.im * b.im) * inv, fma(a.im, b.re, -(a.re * b.im)) * inv};
}
__device__ __forceinline__ cplx csqrt_(cplx z) {
#pragma clang fp contract(off)
  // principal branch
  const double mag = hypot(z.re, z.im);
  double re = sqrt(0.5 * (mag + fabs(z.re)));
  double im = (re == 0.0) ? 0.0 : 0.5 * z.im / re;
  if (z.re < 0.0) {
    const double t = re;
    re = fabs(im);
    im = copysign(t, z.im);
  }
  return {re, im};
}
__device__ __forceinline__ cplx cconj(cplx a) { return {a.re, -a.im}; }
__device__ __forceinline__ cplx cneg(cplx a) { return {-a.re, -a.im}; }
__device__ __forceinline__ cplx cscale(cplx a, double s) {
#pragma clang fp contract(off)
  return {a.re * s, a.im * s};
}

}  // namespace nlc
