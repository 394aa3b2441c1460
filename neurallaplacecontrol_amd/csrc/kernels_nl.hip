// Representation MLP + sphere->complex + Fourier ILT + state update + running cost, fused.
//   LaplaceRepresentationFunc.forward  w_nl.py:55-63
//   torchlaplace.laplace_reconstruct   (external; call site w_nl.py:137-144)  -- Fourier ILT, scale == 2
//   harness dynamics / running_cost    mppi_with_model.py:103-122, 145-171
//   horizon loop                       planners/mppi_delay.py:271-296
//
// One wavefront owns 16 samples for the WHOLE horizon: state, hidden activations and the ILT partial sums
// stay in MFMA accumulator layout (feature on rows/registers, sample on columns/lanes) across all T steps.
// The only HBM traffic per (sample, step) is the GRU latent in (16 B), the action in (nu*8 B) and the
// state out (d*8 B); the (2dS) theta/phi tensor of the reference (22 MB/step at cfg2) never exists.
//
// Layer-3 rows are permuted on the host into "slots": tile j, register r in {0,1} holds theta of ILT
// element e = 4(2j+r)+q, register r+2 holds phi of the same element, so the sphere->complex map
// F = tan(phi/2+pi/4) e^{i theta} is lane-local.  Elements are ordered even-k first, odd-k second, because
// with scale == 2 the Fourier phase e^{i pi k t/T} is i^k: even-k terms need only cos(theta),
// odd-k terms only sin(theta).  The sum over k is one more MFMA with a constant (dims x elements)
// coefficient matrix; its output rows are the state dims, which is exactly the B-fragment layout of the
// next step's layer-1 input, so x <- x + dx needs no lane movement either.
//
// Roofline: FP64 MFMA bound; per 16 samples and step (2 + h/4)*HT + (h/4)*nt3 + 2*nt3 MFMAs.
#include "nlc_nl_kernels.h"

#if NLC_PHASE_CLOCKS
namespace nlc {
__device__ unsigned long long nlc_phase_clk[16];
}
// tools/rollout_phase_clocks.py: reads (and clears) the phase sums of the launches since the last call
extern "C" int nlc_debug_phase_clocks(unsigned long long* out16) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(nlc::nlc_phase_clk), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  const unsigned long long zero[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(nlc::nlc_phase_clk), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif

namespace nlc {

// launchers for hidden width 16 * 8
hipError_t launch_nl_rollout_h128(const RolloutArgs& a, hipStream_t s, bool split) {
  if (split) {
    const unsigned g16 = (unsigned)((a.K + 15) / 16);
    switch (a.net.nt3) {
#define X(N)                                                                                  \
  case N:                                                                                     \
    hipLaunchKernelGGL((nl_rollout_split_kernel<8, N>), dim3(g16), dim3(256), 0, s, a);   \
    break;
      NLC_FOR_NT3(X)
#undef X
      default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  const unsigned grid = (unsigned)((a.K + 63) / 64);
  switch (a.net.nt3) {
#define X(N)                                                                            \
  case N:                                                                               \
    hipLaunchKernelGGL((nl_rollout_kernel<8, N>), dim3(grid), dim3(256), 0, s, a);  \
    break;
    NLC_FOR_NT3(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------ per-step tail of the staged (de Hoog) planner path
// x <- x + dx (mppi_with_model.py:120-121), store, running cost and perturbation cost of horizon step t.  The staged path runs
// this launch for the LAST step only (the tails of the steps before ride in the next representation launch, nlc_rollout.h); it
// lives in this translation unit -- not in kernels_mppi.hip, which is built without a*b+c contraction -- so that every step's
// cost is the same arithmetic (round 4: the persistent step chain is tested bit for bit against this path).
__global__ __launch_bounds__(256) void step_tail_kernel(const StepTailArgs a) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= a.K) return;
  double x[NLC_MAX_D];
  const int64_t e = k / a.Kep;
  const double* src = a.first ? a.state0 + (a.state_per_sample ? k : e) * a.d : a.x + k * a.d;
  for (int i = 0; i < a.d; ++i) {
    x[i] = src[i] + a.dx[k * a.d + i];
    a.x[k * a.d + i] = x[i];
    if (a.states != nullptr) a.states[(k * a.T + a.t) * a.d + i] = x[i];
  }
  double u[NLC_MAX_NU] = {0.0, 0.0};
  for (int j = 0; j < a.nu; ++j) u[j] = a.u_scale * a.perturbed[(k * a.T + a.t) * a.nu + j];
  const double pc = perturbation_cost_step(a.noise + (k * a.T + a.t) * a.nu, a.U + (e * a.T + a.t) * a.nu, a.sigma_inv, a.lambda_,
                                           a.nu, a.noise_abs_cost);
  const double cost = (a.first ? 0.0 : a.ccarry[k * 2]) + running_cost(a.env, x, u, a.nu);
  const double pcost = (a.first ? 0.0 : a.ccarry[k * 2 + 1]) + pc;
  if (a.last) {
    a.cost_total[k] = cost + pcost;
  } else {
    a.ccarry[k * 2] = cost;
    a.ccarry[k * 2 + 1] = pcost;
  }
}
hipError_t launch_step_tail(const StepTailArgs& a, hipStream_t s) {
  if (a.K <= 0) return hipSuccess;
  hipLaunchKernelGGL(step_tail_kernel, dim3((unsigned)((a.K + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}


int nl_pick_nt3(int need) {
#define X(N) \
  if (need <= N) return N;
  NLC_FOR_NT3(X)
#undef X
  return -1;
}

// this width's forward / representation launchers: kernels_nl_fwd.hip, kernels_nl_fwdt.hip, kernels_nl_rep.hip
hipError_t launch_nl_forward_h128_const(const ForwardArgs& a, hipStream_t s);    // kernels_nl_fwd.hip
hipError_t launch_nl_forward_h128_general(const ForwardArgs& a, hipStream_t s);  // kernels_nl_fwdt.hip
hipError_t launch_nl_forward_h128(const ForwardArgs& a, hipStream_t s) {
  return a.const_t ? launch_nl_forward_h128_const(a, s) : launch_nl_forward_h128_general(a, s);
}
hipError_t launch_nl_repfunc_h128(const RepFuncArgs& a, hipStream_t s);
// the other hidden widths: translation units of their own (kernels_nl_h64.hip, kernels_nl_h256*.hip)
hipError_t launch_nl_rollout_h64(const RolloutArgs& a, hipStream_t s, bool split);
hipError_t launch_nl_rollout_h256(const RolloutArgs& a, hipStream_t s, bool split);
hipError_t launch_nl_forward_h64(const ForwardArgs& a, hipStream_t s);
hipError_t launch_nl_forward_h256(const ForwardArgs& a, hipStream_t s);
hipError_t launch_nl_repfunc_h64(const RepFuncArgs& a, hipStream_t s);
hipError_t launch_nl_repfunc_h256(const RepFuncArgs& a, hipStream_t s);

hipError_t launch_nl_rollout_lin_h64(const RolloutArgs& a, hipStream_t s, bool split);
hipError_t launch_nl_rollout_lin_h128(const RolloutArgs& a, hipStream_t s, bool split);
hipError_t launch_nl_rollout_lin_h256(const RolloutArgs& a, hipStream_t s, bool split);
hipError_t launch_nl_rollout(const RolloutArgs& a, hipStream_t s, int force_variant) {
  if (a.K <= 0) return hipSuccess;
  // one wave per 16-sample tile fills the 1024 SIMDs only for K >= 16384; below that, split the tile over
  // the 4 waves of a workgroup (force_variant: 0 auto, 1 wave-per-tile, 2 split)
  const bool split = force_variant == 2 || (force_variant == 0 && a.K <= 8192);
  if (a.net.lin) {
    switch (a.net.h) {
      case 64: return launch_nl_rollout_lin_h64(a, s, split);
      case 128: return launch_nl_rollout_lin_h128(a, s, split);
      case 256: return launch_nl_rollout_lin_h256(a, s, split);
      default: return hipErrorInvalidValue;
    }
  }
  switch (a.net.h) {
    case 64: return launch_nl_rollout_h64(a, s, split);
    case 128: return launch_nl_rollout_h128(a, s, split);
    case 256: return launch_nl_rollout_h256(a, s, split);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_nl_forward(const ForwardArgs& a, hipStream_t s) {
  if (a.N <= 0) return hipSuccess;
  switch (a.net.h) {
    case 64: return launch_nl_forward_h64(a, s);
    case 128: return launch_nl_forward_h128(a, s);
    case 256: return launch_nl_forward_h256(a, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_nl_repfunc(const RepFuncArgs& a, hipStream_t s) {
  if (a.N <= 0) return hipSuccess;
  switch (a.net.h) {
    case 64: return launch_nl_repfunc_h64(a, s);
    case 128: return launch_nl_repfunc_h128(a, s);
    case 256: return launch_nl_repfunc_h256(a, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace nlc

// tools/split_phase_clocks.py (a -DNLC_PHASE_CLOCKS=1 build of this unit): phase sums of the latency-split bodies launched from here
namespace nlc {
NLC_DEFINE_SPLIT_CLK_READER(nlc_debug_split_clocks_nl)
}
