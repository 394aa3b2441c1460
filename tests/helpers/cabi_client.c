/* Plain-C client of include/nlc.h: MPPI commands with oracle cartpole dynamics (the third through the library's own RCCL
 * communicator), one env step, and the Fourier line
 * integral with its backward; device buffers from the HIP runtime's C API, no Python and no torch.  Prints the action, the first cost and beta/eta so the GPU test can
 * compare them with the Python mirror driving the same library (device Philox noise, same seed and counter).
 *   gcc -std=c99 cabi_client.c -I include -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -L<libdir> -lnlc_hip
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,<libdir> -Wl,-rpath,/opt/rocm/lib -lm -o cabi_client */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nlc.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if (rc_ != 0) {                                                              \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, nlc_last_error(ctx));       \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

static double* dev_alloc(size_t n) {
  void* p = NULL;
  if (hipMalloc(&p, n * sizeof(double)) != hipSuccess) exit(2);
  return (double*)p;
}

int main(void) {
  nlc_ctx* ctx = NULL;
  if (nlc_create(0, &ctx) != NLC_OK) {
    fprintf(stderr, "nlc_create: %s\n", nlc_last_error(NULL));
    return 3;
  }
  const int K = 512, T = 10, nu = 1, d = 5, B = 4;
  const double A = 3.0;
  nlc_mppi_desc md;
  memset(&md, 0, sizeof(md));
  md.K = K; md.K_global = K; md.k_offset = 0; md.T = T; md.nu = nu; md.d = d; md.B = B;
  md.lambda_ = 1.0; md.u_scale = A; md.has_bounds = 1; md.u_min[0] = -A; md.u_max[0] = A;
  md.noise_sigma[0] = 1.0; md.noise_sigma_inv[0] = 1.0; md.noise_chol[0] = 1.0;
  md.u_per_command = 1; md.dynamics = NLC_DYN_ORACLE; md.env = NLC_ENV_CARTPOLE; md.delay = 2; md.ts_pred = 0.05;
  CHECK(nlc_mppi_configure(ctx, &md));
  nlc_mppi_buffers buf;
  memset(&buf, 0, sizeof(buf));
  buf.noise = dev_alloc((size_t)K * T * nu);
  buf.perturbed = dev_alloc((size_t)K * T * nu);
  buf.states = dev_alloc((size_t)K * T * d);
  buf.cost_total = dev_alloc(K);
  buf.cost_nz = dev_alloc(K);
  buf.omega = dev_alloc(K);
  buf.partials = dev_alloc(2 + T * nu);
  buf.action = dev_alloc(nu);
  const long long ws = nlc_mppi_workspace_bytes(ctx);
  if (ws < 0 || hipMalloc(&buf.workspace, (size_t)ws) != hipSuccess) return 4;
  double U[10] = {0}, state[5] = {0.01, 0.0, -1.0, 0.02, 0.0}, abuf[4] = {0.5, -0.25, 0.0, 1.0}, action[1] = {0};
  CHECK(nlc_mppi_set_U(ctx, U));
  /* tuning knobs: a known name is accepted, an unknown one is NLC_ERR_BAD_ARG */
  CHECK(nlc_set_option(ctx, "rollout_variant", 0.0));
  if (nlc_set_option(ctx, "no_such_option", 1.0) != NLC_ERR_BAD_ARG) return 6;
  for (int cmd = 0; cmd < 3; ++cmd) {
    CHECK(nlc_mppi_rollout(ctx, state, 0, abuf, &buf, /*rng=*/1, /*seed=*/17, /*counter=*/(uint64_t)cmd));
    if (cmd < 2) {
      CHECK(nlc_mppi_finish(ctx, buf.partials, 1, 0, &buf, action));
    } else {
      /* third command: the library's own RCCL communicator (one rank here) gathers inside nlc_mppi_finish */
      unsigned char uid[NLC_COMM_ID_BYTES];
      if (nlc_comm_unique_id(uid) != NLC_OK) {
        fprintf(stderr, "nlc_comm_unique_id: %s\n", nlc_last_error(NULL));
        return 7;
      }
      CHECK(nlc_comm_init(ctx, 0, 1, uid));
      if (nlc_mppi_finish(ctx, NULL, 2, 0, &buf, action) != NLC_ERR_BAD_ARG) return 8; /* G != communicator's world */
      CHECK(nlc_mppi_finish(ctx, NULL, 1, 0, &buf, action));
      CHECK(nlc_comm_destroy(ctx));
    }
    double part[2], c0;
    if (hipMemcpy(part, buf.partials, sizeof(part), hipMemcpyDeviceToHost) != hipSuccess) return 5;
    if (hipMemcpy(&c0, buf.cost_total, sizeof(c0), hipMemcpyDeviceToHost) != hipSuccess) return 5;
    printf("%.17g %.17g %.17g %.17g\n", action[0], c0, part[0], part[1]);
  }
  CHECK(nlc_mppi_get_U(ctx, U));
  for (int t = 0; t < T; ++t) printf("%.17g%c", U[t], t + 1 < T ? ' ' : '\n');

  /* env side of the loop: two envs, one control step (delay 1, 3-row action buffers) */
  {
    double st[8] = {0.1, -0.2, 3.0, 0.5, -0.3, 0.4, 2.5, -1.0}, ab[6] = {0.5, 1.0, -2.0, 0.25, -0.5, 1.5};
    double act[2] = {2.0, -1.0}, obs[10], rew[2];
    double *st_d = dev_alloc(8), *ab_d = dev_alloc(6), *act_d = dev_alloc(2), *obs_d = dev_alloc(10), *rew_d = dev_alloc(2);
    hipMemcpy(st_d, st, sizeof(st), hipMemcpyHostToDevice);
    hipMemcpy(ab_d, ab, sizeof(ab), hipMemcpyHostToDevice);
    hipMemcpy(act_d, act, sizeof(act), hipMemcpyHostToDevice);
    CHECK(nlc_env_step(ctx, NLC_ENV_CARTPOLE, 0, 0.05, 1, 2, 3, 1, st_d, ab_d, act_d, obs_d, rew_d));
    CHECK(nlc_synchronize(ctx));
    hipMemcpy(obs, obs_d, sizeof(obs), hipMemcpyDeviceToHost);
    hipMemcpy(rew, rew_d, sizeof(rew), hipMemcpyDeviceToHost);
    for (int i = 0; i < 10; ++i) printf("%.17g ", obs[i]);
    printf("%.17g %.17g\n", rew[0], rew[1]);
  }
  /* torchlaplace.laplace_reconstruct line integral and its backward: 3 points, d = 2, S = 17 */
  {
    enum { N = 3, D = 2, S = 17 };
    double th[N * D * S], ph[N * D * S], tt[N] = {0.1, 0.125, 0.3}, gx[N * D], x[N * D], gth[N * D * S], gph[N * D * S];
    for (int i = 0; i < N * D * S; ++i) {
      th[i] = 3.0 * ((i * 37) % 101) / 101.0 - 1.5;
      ph[i] = 1.2 * ((i * 53) % 97) / 97.0 - 0.6;
    }
    for (int i = 0; i < N * D; ++i) gx[i] = 1.0 + 0.5 * i;
    double *th_d = dev_alloc(N * D * S), *ph_d = dev_alloc(N * D * S), *t_d = dev_alloc(N), *gx_d = dev_alloc(N * D);
    double *x_d = dev_alloc(N * D), *gth_d = dev_alloc(N * D * S), *gph_d = dev_alloc(N * D * S);
    hipMemcpy(th_d, th, sizeof(th), hipMemcpyHostToDevice);
    hipMemcpy(ph_d, ph, sizeof(ph), hipMemcpyHostToDevice);
    hipMemcpy(t_d, tt, sizeof(tt), hipMemcpyHostToDevice);
    hipMemcpy(gx_d, gx, sizeof(gx), hipMemcpyHostToDevice);
    /* one line per algorithm: fourier, dehoog, fixed_tablot (17 terms each; defaults of the Python mirror's ilt_desc) */
    const int algos[3] = {NLC_ILT_FOURIER, NLC_ILT_DEHOOG, NLC_ILT_FIXED_TALBOT};
    const double alphas[3] = {1e-3, 1e-10, 1.0}, tols[3] = {1e-2, 1e-9, 10.0}, scales[3] = {2.0, 2.0, 1.0};
    for (int m = 0; m < 3; ++m) {
      nlc_ilt_desc il;
      il.algo = algos[m]; il.terms = S; il.alpha = alphas[m]; il.tol = tols[m]; il.scale = scales[m];
      CHECK(nlc_ilt_reconstruct(ctx, &il, th_d, ph_d, t_d, N, D, x_d));
      CHECK(nlc_ilt_reconstruct_backward(ctx, &il, th_d, ph_d, t_d, gx_d, N, D, gth_d, gph_d));
      CHECK(nlc_synchronize(ctx));
      hipMemcpy(x, x_d, sizeof(x), hipMemcpyDeviceToHost);
      hipMemcpy(gth, gth_d, sizeof(gth), hipMemcpyDeviceToHost);
      hipMemcpy(gph, gph_d, sizeof(gph), hipMemcpyDeviceToHost);
      double s1 = 0.0, s2 = 0.0;
      for (int i = 0; i < N * D * S; ++i) { s1 += gth[i] * (1 + i % 3); s2 += gph[i] * (1 + i % 5); }
      for (int i = 0; i < N * D; ++i) printf("%.17g ", x[i]);
      printf("%.17g %.17g\n", s1, s2);
    }
  }
  nlc_destroy(ctx);
  return 0;
}
