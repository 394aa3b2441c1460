// LIN instances of the rollout kernels, hidden width 128 (the harness's hidden_units): fixed Talbot / Stehfest models on the
// fused rollout (round 3).  Their reconstruction x = (1/t) sum_k (w_re,k Re F_k - w_im,k Im F_k) is linear in F like the Fourier
// sum, so it rides the MFMA epilogue too -- with BOTH components of F = R e^{i theta} per term: two ILT MFMAs per slot group
// against the coefficient fragments NlNetArgs::Cp (w_re / t) and Cp2 (-w_im / t), which nlc_mppi_configure folds for the
// planner's constant prediction time.  A translation unit of its own: the Fourier instances keep their code and registers.
#include "nlc_nl_kernels.h"

namespace nlc {

hipError_t launch_nl_rollout_lin_h128(const RolloutArgs& a, hipStream_t s, bool split) {
  if (a.net.Cp2 == nullptr || a.net.lin != 1) return hipErrorInvalidValue;
  if (split) {
    const unsigned g16 = (unsigned)((a.K + 15) / 16);
    switch (a.net.nt3) {
#define X(N)                                                                                      \
  case N:                                                                                         \
    hipLaunchKernelGGL((nl_rollout_split_kernel<8, N, true>), dim3(g16), dim3(256), 0, s, a);   \
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
#define X(N)                                                                                \
  case N:                                                                                   \
    hipLaunchKernelGGL((nl_rollout_kernel<8, N, true>), dim3(grid), dim3(256), 0, s, a);  \
    break;
    NLC_FOR_NT3(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace nlc
