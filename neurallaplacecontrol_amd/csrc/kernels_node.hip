// NODE baseline dynamics (train_utils.py:637-738) on FP64 matrix cores: the ODE function
//   f(x, u) = Linear(d+aug+nu, H) -> tanh -> Linear(H, H) -> tanh -> Linear(H, d+aug)      (xOdeFuncInXAndU, :637-661)
// integrated by torchdiffeq's fixed-grid Euler solver over [0, ts_pred / (dt*8)] with step_size 0.05 (:717-723; the
// solver is restated -- parity unpinned for odeint, oracle/node_model.py), inside the planner's horizon loop
// (state + model(state, window, ts_pred), mppi_with_model.py:103-122).
//
// Same dataflow as the NL rollout (kernels_nl.hip): one wavefront per 16-sample tile, weights are the MFMA A operand
// streamed fragment-packed from L2, samples are the B/D columns, accumulator register r of output tile j IS the B
// fragment of k-step 4j+r of the next layer.  The ODE state y (d + aug <= 8 values) lives in two registers per lane
// in exactly the accumulator layout of the last layer's output tile (row 4r + q), so an Euler sub-step is
// y_r += h * o_r with no lane movement, and y_r is directly the layer-1 B fragment of the next sub-step.
//
// Hidden activations: the short tanh pair of nlc_math.h (tanh_pair_fast: absolute error <= 1.5e-13, one reciprocal per
// pair; round 3: 6.39 -> 6.31 ms per K = 16384, T = 40 rollout, every test at its tolerance).
// Roofline: FP64 MFMA.  Per 16 samples and Euler sub-step: 3 HT + 4 HT^2 + 4 HT MFMAs (HT = ceil(H/16) = 17 for
// H = 270: 1275 MFMAs of 2048 flop); three sub-steps per horizon step at the harness's dt.
#include "nlc_device.h"
#include "nlc_envcost.h"
#include "nlc_kernels.h"

namespace nlc {

// One evaluation of the ODE function for the wave's 16 samples.  in0/in1/in2: layer-1 B fragments (input index 4s+q
// of [y (d+aug) | u (nu)]).  Returns the output tile: register r of lane group q = row 4r + q of f.
template <int HT>
__device__ __forceinline__ v4d node_eval(const NodeNetArgs& n, int lane, int q, double in0, double in1, double in2) {
  constexpr int KS = HT * 4;
  v4d h1[HT];
#pragma unroll
  for (int j = 0; j < HT; ++j) h1[j] = load_bias_tile(n.b1, j, q);
  gemm_acc<HT, 3>(h1, n.W1p, lane, [&](int ks) { return ks == 0 ? in0 : (ks == 1 ? in1 : in2); });
#pragma unroll
  for (int j = 0; j < HT; ++j)
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      double ta, tb;
      m::tanh_pair_fast(h1[j][r], h1[j][r + 1], &ta, &tb);
      h1[j][r] = ta;
      h1[j][r + 1] = tb;
    }
  v4d h2[HT];
#pragma unroll
  for (int j = 0; j < HT; ++j) h2[j] = load_bias_tile(n.b2, j, q);
  gemm_acc<HT, KS>(h2, n.W2p, lane, [&](int ks) { return h1[ks >> 2][ks & 3]; });
#pragma unroll
  for (int j = 0; j < HT; ++j)
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      double ta, tb;
      m::tanh_pair_fast(h2[j][r], h2[j][r + 1], &ta, &tb);
      h2[j][r] = ta;
      h2[j][r + 1] = tb;
    }
  v4d o[1];
  o[0] = load_bias_tile(n.b3, 0, q);
  gemm_acc<1, KS>(o, n.W3p, lane, [&](int ks) { return h2[ks >> 2][ks & 3]; });
  return o[0];
}

// Euler integration of the augmented normalised state for the wave's samples.  y0/y1: rows q and 4+q of
// [x_norm | aug]; ua/ub/uc: this lane's u entries for input indices q, 4+q, 8+q (0 where the index is not an action).
template <int HT>
__device__ __forceinline__ void node_integrate(const NodeNetArgs& n, int lane, int q, double& y0, double& y1, double ua,
                                               double ub, double uc) {
  const int dy = n.d + n.aug;
  for (int s = 0; s < n.nsub; ++s) {
    const double in0 = (q < dy) ? y0 : ua;
    const double in1 = (4 + q < dy) ? y1 : ub;
    const v4d o = node_eval<HT>(n, lane, q, in0, in1, uc);
    const double h = n.hsub[s];
    if (q < dy) y0 = y0 + h * o[0];
    if (4 + q < dy) y1 = y1 + h * o[1];
  }
}

template <int HT>
__global__ __launch_bounds__(256) void node_rollout_kernel(const NodeRolloutArgs a) {
  const NodeNetArgs& n = a.net;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, c = lane & 15;
  const int64_t k = ((int64_t)blockIdx.x * 4 + wave) * 16 + c;
  const bool valid = k < a.K;
  const int64_t kc = valid ? k : a.K - 1;
  const int d = n.d, dy = n.d + n.aug;
  const int i0 = q, i1 = 4 + q, i2 = 8 + q;  // input indices this lane feeds
  double x0 = 0.0, x1 = 0.0, m0 = 0.0, m1 = 0.0, s0 = 1.0, s1 = 1.0;
  const int ep = (int)(kc / a.Kep);
  const int uoff = ep * a.T * a.nu;
  const double* st = a.state0 + (a.state_per_sample ? kc : (int64_t)ep) * d;
  if (i0 < d) {
    x0 = st[i0];
    m0 = n.state_mean[i0];
    s0 = n.state_std[i0];
  }
  if (i1 < d) {
    x1 = st[i1];
    m1 = n.state_mean[i1];
    s1 = n.state_std[i1];
  }
  double cost = 0.0, pcost = 0.0;
  for (int t = 0; t < a.T; ++t) {
    double u[NLC_MAX_NU] = {0.0, 0.0};
    // the model reads window[:, -1, :] = the newest action of the window = u_scale * perturbed[k, t], RAW (:715)
    for (int j = 0; j < a.nu; ++j) u[j] = a.u_scale * a.perturbed[(kc * a.T + t) * a.nu + j];
    auto upick = [&](int i) {
      const int j = i - dy;
      return (j == 0) ? u[0] : ((j == 1 && a.nu > 1) ? u[1] : 0.0);
    };
    double y0 = (i0 < d) ? (x0 - m0) / s0 : 0.0;  // augmented rows start at zero (:705-708)
    double y1 = (i1 < d) ? (x1 - m1) / s1 : 0.0;
    node_integrate<HT>(n, lane, q, y0, y1, upick(i0), upick(i1), upick(i2));
    // state + model(state, window, ts_pred): the model's output is the INTEGRATED normalised state (:724)
    if (i0 < d) x0 = x0 + y0;
    if (i1 < d) x1 = x1 + y1;
    if (valid && a.states != nullptr) {
      double* so = a.states + (k * a.T + t) * d;
      if (i0 < d) so[i0] = x0;
      if (i1 < d) so[i1] = x1;
    }
    double xs[NLC_MAX_D];
#pragma unroll
    for (int i = 0; i < NLC_MAX_D; ++i) xs[i] = __shfl((i < 4) ? x0 : x1, ((i & 3) << 4) | c, 64);
    double pc = 0.0;
    for (int j = 0; j < a.nu; ++j) {
      double acj = 0.0;
      for (int i = 0; i < a.nu; ++i) {
        double e = a.noise[(kc * a.T + t) * a.nu + i];
        if (a.noise_abs_cost) e = fabs(e);
        acj += (a.lambda_ * e) * a.sigma_inv[i * a.nu + j];
      }
      pc += a.U[uoff + t * a.nu + j] * acj;
    }
    cost += running_cost(a.env, xs, u, a.nu);
    pcost += pc;
  }
  if (valid && q == 0) a.cost_total[k] = cost + pcost;
}

// NODE.forward: obs (N, d), newest action (N, nu) -> integrated normalised state (N, d)
template <int HT>
__global__ __launch_bounds__(256) void node_forward_kernel(const NodeForwardArgs a) {
  const NodeNetArgs& n = a.net;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, c = lane & 15;
  const int64_t r = ((int64_t)blockIdx.x * 4 + wave) * 16 + c;
  const bool valid = r < a.N;
  const int64_t rc = valid ? r : a.N - 1;
  const int d = n.d, dy = n.d + n.aug;
  const int i0 = q, i1 = 4 + q, i2 = 8 + q;
  double y0 = (i0 < d) ? (a.obs[rc * d + i0] - n.state_mean[i0]) / n.state_std[i0] : 0.0;
  double y1 = (i1 < d) ? (a.obs[rc * d + i1] - n.state_mean[i1]) / n.state_std[i1] : 0.0;
  auto upick = [&](int i) {
    const int j = i - dy;
    return (j >= 0 && j < n.nu) ? a.action[rc * n.nu + j] : 0.0;
  };
  node_integrate<HT>(n, lane, q, y0, y1, upick(i0), upick(i1), upick(i2));
  if (valid) {
    if (i0 < d) a.out[r * d + i0] = y0;
    if (i1 < d) a.out[r * d + i1] = y1;
  }
}

hipError_t launch_node_rollout(const NodeRolloutArgs& a, int ht, hipStream_t s) {
  if (a.K <= 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.K + 63) / 64);
  switch (ht) {
    case 4: hipLaunchKernelGGL((node_rollout_kernel<4>), dim3(grid), dim3(256), 0, s, a); break;
    case 8: hipLaunchKernelGGL((node_rollout_kernel<8>), dim3(grid), dim3(256), 0, s, a); break;
    case 17: hipLaunchKernelGGL((node_rollout_kernel<17>), dim3(grid), dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_node_forward(const NodeForwardArgs& a, int ht, hipStream_t s) {
  if (a.N <= 0) return hipSuccess;
  const unsigned grid = (unsigned)((a.N + 63) / 64);
  switch (ht) {
    case 4: hipLaunchKernelGGL((node_forward_kernel<4>), dim3(grid), dim3(256), 0, s, a); break;
    case 8: hipLaunchKernelGGL((node_forward_kernel<8>), dim3(grid), dim3(256), 0, s, a); break;
    case 17: hipLaunchKernelGGL((node_forward_kernel<17>), dim3(grid), dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace nlc
