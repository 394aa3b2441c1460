// Host side of libnlc_hip.so, ILT unit: the stand-alone pieces of torchlaplace.laplace_reconstruct (query points + sphere
// projection, reconstruction, backward) and the node / weight tables of the linear algorithms.
#include "nlc_host.h"

using namespace nlc;
using namespace nlc::host;

namespace nlc {
namespace host {

void sphere_inputs(const nlc_ilt_desc& ilt, double tn, std::vector<double>& sph) {
  // [theta_s(0..S-1) | phi_s(0..S-1)] of s_k = gamma + i pi k / T (Fourier, de Hoog) or s_k = node_k / t (linear algorithms)
  const int S = ilt.terms;
  sph.assign(2 * S, 0.0);
  if (ilt.algo == NLC_ILT_FIXED_TALBOT || ilt.algo == NLC_ILT_STEHFEST) {
    std::vector<double> tab;
    linear_tables_host(ilt.algo, S, tab);
    for (int k = 0; k < S; ++k) {
      const double re = tab[k] / tn, im = tab[S + k] / tn;
      sph[k] = std::atan2(im, re);
      const double a2 = re * re + im * im;
      sph[S + k] = std::asin((a2 - 1.0) / (a2 + 1.0));
    }
    return;
  }
  const double Tt = ilt.scale * tn;
  const double gamma = ilt.alpha - std::log(ilt.tol) / (ilt.scale * Tt);
  for (int k = 0; k < S; ++k) {
    const double im = M_PI * (double)k / Tt;
    sph[k] = std::atan2(im, gamma);
    const double a2 = gamma * gamma + im * im;
    sph[S + k] = std::asin((a2 - 1.0) / (a2 + 1.0));
  }
}

int check_ilt(nlc_ctx* c, const nlc_ilt_desc* d) {
  if (!d) return fail(c, NLC_ERR_BAD_ARG, "ilt desc is NULL");
  if (d->terms < 1 || d->terms > kMaxTerms) return fail(c, NLC_ERR_BAD_SHAPE, "ilt terms out of range [1,129]");
  if (!(d->tol > 0.0) || !(d->scale > 0.0)) return fail(c, NLC_ERR_BAD_ARG, "ilt tol/scale must be positive");
  if (d->algo != NLC_ILT_FOURIER && d->algo != NLC_ILT_DEHOOG && d->algo != NLC_ILT_FIXED_TALBOT &&
      d->algo != NLC_ILT_STEHFEST)
    return fail(c, NLC_ERR_UNSUPPORTED, "ilt_algorithm: fourier, dehoog, fixed_tablot and stehfest are implemented");
  if (d->algo == NLC_ILT_STEHFEST && (d->terms % 2 != 0 || d->terms < 2 || d->terms > 20))
    return fail(c, NLC_ERR_UNSUPPORTED, "stehfest: ilt_reconstruction_terms must be even, 2 .. 20 (Salzer weights in float64)");
  if (d->algo == NLC_ILT_FIXED_TALBOT && d->terms < 2)
    return fail(c, NLC_ERR_UNSUPPORTED, "fixed_tablot: ilt_reconstruction_terms must be >= 2");
  return NLC_OK;
}

// nodes and weights of the linear algorithms (mpmath 1.3.0 calculus/inverselaplace.py: FixedTalbot.calc_laplace_parameter /
// calc_time_domain_solution, Stehfest._coeff), uploaded once per (algorithm, terms): [node_re | node_im | w_re | w_im]
void linear_tables_host(int algo, int S, std::vector<double>& h) {
  h.assign((size_t)4 * S, 0.0);
  double *nr = h.data(), *ni = nr + S, *wr = ni + S, *wi = wr + S;
  if (algo == NLC_ILT_FIXED_TALBOT) {
    const int M = S;
    const double r = 2.0 * M / 5.0;
    nr[0] = r;
    wr[0] = 0.4 * std::exp(r) / 2.0;
    for (int k = 1; k < M; ++k) {
      const double th = k * M_PI / M, cot = 1.0 / std::tan(th);
      nr[k] = r * th * cot;
      ni[k] = r * th;
      const double e = 0.4 * std::exp(nr[k]), cr = std::cos(ni[k]), ci = std::sin(ni[k]);
      const double fi = th * (1.0 + cot * cot) - cot;  // factor 1 + i fi
      wr[k] = e * (cr - ci * fi);
      wi[k] = e * (ci + cr * fi);
    }
  } else {
    const int M = S, M2 = S / 2;
    auto fac = [](int n) {
      long double f = 1.0L;
      for (int i = 2; i <= n; ++i) f *= i;
      return f;
    };
    for (int k = 1; k <= M; ++k) {
      long double z = 0.0L;
      for (int j = (k + 1) / 2; j <= (k < M2 ? k : M2); ++j)
        z += std::pow((long double)j, M2) * fac(2 * j) / (fac(M2 - j) * fac(j) * fac(j - 1) * fac(k - j) * fac(2 * j - k));
      nr[k - 1] = k * M_LN2;
      wr[k - 1] = (double)(((k + M2) % 2 ? -1.0L : 1.0L) * z * (long double)M_LN2);
    }
  }
}
int linear_tables(nlc_ctx* c, const nlc_ilt_desc* d, const double** tab) {
  const int S = d->terms;
  if (c->lin_tab && c->lin_algo == d->algo && c->lin_S == S) {
    *tab = c->lin_tab;
    return NLC_OK;
  }
  std::vector<double> h;
  linear_tables_host(d->algo, S, h);
  if (!c->lin_tab) NLC_HIP(c, hipMalloc((void**)&c->lin_tab, (size_t)4 * kMaxTerms * sizeof(double)));
  // a kernel of an earlier call may still read the old tables -- on ANY stream the ctx was bound to since (the Python
  // mirror rebinds it to torch's current stream every call): this rare path waits for the whole device
  NLC_HIP(c, hipDeviceSynchronize());
  NLC_HIP(c, hipMemcpy(c->lin_tab, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
  c->lin_algo = d->algo;
  c->lin_S = S;
  *tab = c->lin_tab;
  return NLC_OK;
}

}  // namespace host
}  // namespace nlc

extern "C" int nlc_ilt_rep_inputs(nlc_ctx* c, const nlc_ilt_desc* d, const double* p, const double* t, int t_batched,
                                  int64_t B, int64_t Tt, int P, double* out) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (int r = check_ilt(c, d)) return r;
  if (B < 0 || Tt < 0 || P < 0) return fail(c, NLC_ERR_BAD_SHAPE, "negative shape");
  if (B * Tt == 0) return NLC_OK;
  if (!p || !t || !out) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  RepInArgs a{p, t, out, B, Tt, P, d->terms, t_batched, d->alpha, std::log(d->tol), d->scale, nullptr, nullptr, 1.0};
  if (d->algo == NLC_ILT_FIXED_TALBOT || d->algo == NLC_ILT_STEHFEST) {
    const double* tab = nullptr;
    if (int r = linear_tables(c, d, &tab)) return r;
    a.node_re = tab;
    a.node_im = tab + d->terms;
  }
  ProfScope ps(c, "rep_inputs_kernel");
  NLC_HIP(c, launch_rep_inputs(a, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_ilt_reconstruct(nlc_ctx* c, const nlc_ilt_desc* d, const double* theta, const double* phi,
                                   const double* t, int64_t N, int dd, double* x) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (int r = check_ilt(c, d)) return r;
  if (N < 0 || dd < 1) return fail(c, NLC_ERR_BAD_SHAPE, "bad N or d");
  if (N == 0) return NLC_OK;
  if (!theta || !phi || !t || !x) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  IltArgs a{theta, phi, t, x, N, dd, d->terms, d->alpha, std::log(d->tol), d->scale, nullptr, nullptr, 1.0, 1, 0, 0, 0};
  if (d->algo == NLC_ILT_FIXED_TALBOT || d->algo == NLC_ILT_STEHFEST) {
    const double* tab = nullptr;
    if (int r = linear_tables(c, d, &tab)) return r;
    // the Fourier kernel's coalesced stream with the algorithm's per-term phase and weight (round 3); the one-thread-per-row
    // kernel remains for a term count the stream's tiling does not take
    a.lin_wr = tab + 2 * d->terms;
    a.lin_wi = tab + 3 * d->terms;
    hipError_t le;
    {
      ProfScope ps(c, "ilt_linear_stream_kernel");
      le = launch_ilt_fourier(a, c->stream);
    }
    if (le == hipErrorInvalidValue) {
      (void)hipGetLastError();
      IltLinArgs la{theta, phi, t, x, N, dd, d->terms, a.lin_wr, a.lin_wi};
      ProfScope ps(c, "ilt_linear_kernel");
      NLC_HIP(c, launch_ilt_linear(la, c->stream));
    } else {
      NLC_HIP(c, le);
    }
  } else if (d->algo == NLC_ILT_FOURIER) {
    ProfScope ps(c, "ilt_fourier_kernel");
    NLC_HIP(c, launch_ilt_fourier(a, c->stream));
  } else {
    if (d->terms < 3 || d->terms > 33 || d->terms % 2 == 0)
      return fail(c, NLC_ERR_UNSUPPORTED, "dehoog: ilt_reconstruction_terms must be odd, 3 .. 33 (2M+1 terms)");
    ProfScope ps(c, "ilt_dehoog_kernel");
    NLC_HIP(c, launch_ilt_dehoog(a, c->stream));
  }
  return NLC_OK;
  NLC_GUARD_END(c)
}

extern "C" int nlc_ilt_reconstruct_backward(nlc_ctx* c, const nlc_ilt_desc* d, const double* theta, const double* phi,
                                            const double* t, const double* grad_x, int64_t N, int dd,
                                            double* grad_theta, double* grad_phi) {
  if (!c) return NLC_ERR_BAD_ARG;
  NLC_GUARD_BEGIN
  if (int r = check_ilt(c, d)) return r;
  if (d->algo == NLC_ILT_DEHOOG && (d->terms < 3 || d->terms > 33 || d->terms % 2 == 0))
    return fail(c, NLC_ERR_UNSUPPORTED, "dehoog: ilt_reconstruction_terms must be odd, 3 .. 33 (2M+1 terms)");
  if (N < 0 || dd < 1) return fail(c, NLC_ERR_BAD_SHAPE, "bad N or d");
  if (N == 0) return NLC_OK;
  if (!theta || !phi || !t || !grad_x || !grad_theta || !grad_phi) return fail(c, NLC_ERR_BAD_ARG, "NULL device pointer");
  NLC_HIP(c, hipSetDevice(c->device));
  if (d->algo == NLC_ILT_DEHOOG) {
    // the QD tape lives in stream-ordered scratch of this launch (no ctx state: calls on different streams do not share it)
    const int64_t bytes = ilt_dehoog_bwd_scratch_bytes(N, dd, d->terms, nullptr);
    void* scratch = nullptr;
    NLC_HIP(c, hipMallocAsync(&scratch, (size_t)bytes, c->stream));
    IltDehoogBwdArgs da{theta, phi, t, grad_x, grad_theta, grad_phi, N, dd, d->terms, d->alpha, std::log(d->tol), d->scale, 1.0, scratch};
    hipError_t le;
    {
      ProfScope ps(c, "ilt_dehoog_bwd_kernel");
      le = launch_ilt_dehoog_bwd(da, c->stream);
    }
    const hipError_t fe = hipFreeAsync(scratch, c->stream);
    NLC_HIP(c, le);
    NLC_HIP(c, fe);
    return NLC_OK;
  }
  if (d->algo == NLC_ILT_FIXED_TALBOT || d->algo == NLC_ILT_STEHFEST) {
    const double* tab = nullptr;
    if (int r = linear_tables(c, d, &tab)) return r;
    IltLinBwdArgs la{theta, phi, t, grad_x, grad_theta, grad_phi, N, dd, d->terms, tab + 2 * d->terms, tab + 3 * d->terms};
    ProfScope ps(c, "ilt_linear_bwd_kernel");
    NLC_HIP(c, launch_ilt_linear_bwd(la, c->stream));
    return NLC_OK;
  }
  IltBwdArgs a{theta, phi, t, grad_x, grad_theta, grad_phi, N, dd, d->terms, d->alpha, std::log(d->tol), d->scale, 0, 0};
  ProfScope ps(c, "ilt_fourier_bwd_kernel");
  NLC_HIP(c, launch_ilt_fourier_bwd(a, c->stream));
  return NLC_OK;
  NLC_GUARD_END(c)
}
