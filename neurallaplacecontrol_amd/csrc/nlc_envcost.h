// Env running costs on the trig observation, shared by the MFMA rollout kernels (kernels_nl.hip, kernels_node.hip).
// (kernels_mppi.hip keeps its own copy: that file is built with -ffp-contract=off to follow the torch-CPU op order.)
#pragma once
#include "nlc_device.h"
#include "../../include/nlc.h"

namespace nlc {

// ------------------------------------------------------------------ env running costs (a10)
// cost = -(diff_obs_reward_(x, exp_reward=False) + diff_ac_reward_(u))   mppi_with_model.py:163-164
__device__ __forceinline__ double trig2angle(double c, double s) {
  // base_env.py:297-301: divide by C twice, then atan2
  const double C = c * c + s * s;
  c = c / C;
  s = s / C;
  return atan2(s / C, c / C);
}

__device__ __forceinline__ double running_cost(int env, const double (&x)[NLC_MAX_D], const double (&u)[NLC_MAX_NU],
                                               int nu) {
  // a*b+c is a multiply and an add here, as in the torch-CPU op order (and the same bits in every kernel that inlines this)
#pragma clang fp contract(off)
  if (env < 0) return 0.0;  // cost_external: the caller evaluates its own running cost on the stored states
  double uu = 0.0;
  for (int j = 0; j < nu; ++j) uu += u[j] * u[j];
  if (env == NLC_ENV_CARTPOLE) {
    // ctcartpole.py:303-339,345-346: ee = (x + sin_l, cos_l), goal (0, 1)
    const double e0 = x[0] + x[3] - 0.0, e1 = x[2] - 1.0;
    const double state_reward = -(e0 * e0 + e1 * e1);
    const double vel_reward = -(x[1] * x[1]) - x[4] * x[4];
    return -((state_reward + 0.01 * vel_reward) + (-0.01 * uu));
  } else if (env == NLC_ENV_CARTPOLE_NOTRIG) {
    // ctcartpole.py:297-300 (s.shape[-1] == 4: explicit angle), then :303-339 as above
    const double cl = 1.0 * cos(x[2]), sl = 1.0 * sin(x[2]);
    const double e0 = x[0] + sl - 0.0, e1 = cl - 1.0;
    const double state_reward = -(e0 * e0 + e1 * e1);
    const double vel_reward = -(x[1] * x[1]) - x[3] * x[3];
    return -((state_reward + 0.01 * vel_reward) + (-0.01 * uu));
  } else if (env == NLC_ENV_PENDULUM) {
    // ctpendulum.py:139-155
    const double om = 1.0 - x[0];
    const double state_reward = -(om * om + x[1] * x[1]);
    const double vel_reward = -(x[2] * x[2]);
    return -((state_reward + 0.01 * vel_reward) + (-0.01 * uu));
  } else {
    // ctacrobot.py:233-255 (consts :110-111)
    const double th1 = trig2angle(x[0], x[1]), th2 = trig2angle(x[2], x[3]);
    const double vel_reward = -(x[4] * x[4]) - x[5] * x[5];
    const double p1x = -cos(th1), p1y = sin(th1);
    const double p2x = p1x - cos(th1 + th2), p2y = p1y + sin(th1 + th2);
    const double ex = p2x - 1.0 - 1.0;
    const double state_reward = -(ex * ex) - p2y * p2y;
    return -((state_reward + 1e-1 * vel_reward) + (-1e-4 * uu));
  }
}

// perturbation cost of one horizon step: sum_j U[t,j] (lambda eps Sigma^-1)[j]   (planners/mppi_delay.py:335, 343).
// eps / U: the step's nu entries.  Used by every tail of the de Hoog planner path (staged launches and the persistent chain).
__device__ __forceinline__ double perturbation_cost_step(const double* eps, const double* U, const double* sigma_inv,
                                                         double lambda_, int nu, int noise_abs_cost) {
#pragma clang fp contract(off)
  double pc = 0.0;
  for (int j = 0; j < nu; ++j) {
    double acj = 0.0;
    for (int ii = 0; ii < nu; ++ii) {
      double ev = eps[ii];
      if (noise_abs_cost) ev = fabs(ev);
      acj += (lambda_ * ev) * sigma_inv[ii * nu + j];
    }
    pc += U[j] * acj;
  }
  return pc;
}

}  // namespace nlc
