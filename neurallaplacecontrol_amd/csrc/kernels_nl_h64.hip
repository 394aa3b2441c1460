// Representation MLP + ILT + rollout kernels for hidden_units = 64 (the class default, w_nl.py:72): see kernels_nl.hip.
#include "nlc_nl_kernels.h"

namespace nlc {

// launchers for hidden width 16 * 4
hipError_t launch_nl_rollout_h64(const RolloutArgs& a, hipStream_t s, bool split) {
  if (split) {
    const unsigned g16 = (unsigned)((a.K + 15) / 16);
    switch (a.net.nt3) {
#define X(N)                                                                                  \
  case N:                                                                                     \
    hipLaunchKernelGGL((nl_rollout_split_kernel<4, N>), dim3(g16), dim3(256), 0, s, a);   \
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
    hipLaunchKernelGGL((nl_rollout_kernel<4, N>), dim3(grid), dim3(256), 0, s, a);  \
    break;
    NLC_FOR_NT3(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_nl_forward_h64(const ForwardArgs& a, hipStream_t s) {
  const unsigned grid = (unsigned)((a.N + 63) / 64);
  switch (a.net.nt3) {
#define X(N)                                                                            \
  case N:                                                                               \
    if (a.const_t) {                                                                        \
      hipLaunchKernelGGL((nl_forward_kernel<4, N, false>), dim3(grid), dim3(256), 0, s, a); \
    } else {                                                                                \
      hipLaunchKernelGGL((nl_forward_kernel<4, N, true>), dim3(grid), dim3(256), 0, s, a);  \
    }                                                                                       \
    break;
    NLC_FOR_NT3(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_nl_repfunc_h64(const RepFuncArgs& a, hipStream_t s) {
  const unsigned grid = (unsigned)((a.N + 63) / 64);
  switch (a.net.nt3) {
#define X(N)                                                                                          \
  case N:                                                                                             \
    if (a.general_t) {                                                                                \
      hipLaunchKernelGGL((nl_repfunc_kernel<4, N, true>), dim3(grid), dim3(256), 0, s, a);        \
    } else {                                                                                          \
      hipLaunchKernelGGL((nl_repfunc_kernel<4, N, false>), dim3(grid), dim3(256), 0, s, a);       \
    }                                                                                                 \
    break;
    NLC_FOR_NT3(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace nlc
