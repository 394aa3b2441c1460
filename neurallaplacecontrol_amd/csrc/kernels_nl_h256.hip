// Representation MLP + ILT + rollout kernels for hidden_units = 256: see kernels_nl.hip.
#include "nlc_nl_kernels.h"

namespace nlc {

// launchers for hidden width 16 * 16
hipError_t launch_nl_rollout_h256(const RolloutArgs& a, hipStream_t s, bool split) {
  if (split) {
    const unsigned g16 = (unsigned)((a.K + 15) / 16);
    switch (a.net.nt3) {
#define X(N)                                                                                  \
  case N:                                                                                     \
    hipLaunchKernelGGL((nl_rollout_split_kernel<16, N>), dim3(g16), dim3(256), 0, s, a);   \
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
    hipLaunchKernelGGL((nl_rollout_kernel<16, N>), dim3(grid), dim3(256), 0, s, a);  \
    break;
    NLC_FOR_NT3(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_nl_forward_h256_const(const ForwardArgs& a, hipStream_t s);    // kernels_nl_h256_fwd.hip
hipError_t launch_nl_forward_h256_general(const ForwardArgs& a, hipStream_t s);  // kernels_nl_h256_fwdt.hip
hipError_t launch_nl_forward_h256(const ForwardArgs& a, hipStream_t s) {
  return a.const_t ? launch_nl_forward_h256_const(a, s) : launch_nl_forward_h256_general(a, s);
}

}  // namespace nlc
