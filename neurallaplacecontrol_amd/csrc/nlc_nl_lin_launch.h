// Launcher body of the LIN instances of the rollout kernels for ONE hidden width (kernels_nl_lin*.hip, one translation unit
// per width so they compile in parallel).
#pragma once
#include "nlc_nl_kernels.h"

#define NLC_DEFINE_LIN_ROLLOUT_LAUNCHER(SUFFIX, HT_)                                                     \
  hipError_t launch_nl_rollout_lin_##SUFFIX(const RolloutArgs& a, hipStream_t s, bool split) {          \
    if (a.net.Cp2 == nullptr || a.net.lin != 1) return hipErrorInvalidValue;                            \
    if (split) {                                                                                         \
      const unsigned g16 = (unsigned)((a.K + 15) / 16);                                                  \
      switch (a.net.nt3) {                                                                               \
        NLC_LIN_CASES(nl_rollout_split_kernel, HT_, g16)                                                 \
        default:                                                                                         \
          return hipErrorInvalidValue;                                                                   \
      }                                                                                                  \
      return hipGetLastError();                                                                          \
    }                                                                                                    \
    const unsigned grid = (unsigned)((a.K + 63) / 64);                                                   \
    switch (a.net.nt3) {                                                                                 \
      NLC_LIN_CASES(nl_rollout_kernel, HT_, grid)                                                        \
      default:                                                                                           \
        return hipErrorInvalidValue;                                                                     \
    }                                                                                                    \
    return hipGetLastError();                                                                            \
  }
#define NLC_LIN_CASE(KERNEL, HT_, N, GRID)                                                   \
  case N:                                                                                    \
    hipLaunchKernelGGL((KERNEL<HT_, N, true>), dim3(GRID), dim3(256), 0, s, a);             \
    break;
#define NLC_LIN_CASES(KERNEL, HT_, GRID)                                                                              \
  NLC_LIN_CASE(KERNEL, HT_, 7, GRID) NLC_LIN_CASE(KERNEL, HT_, 9, GRID) NLC_LIN_CASE(KERNEL, HT_, 11, GRID)          \
  NLC_LIN_CASE(KERNEL, HT_, 13, GRID) NLC_LIN_CASE(KERNEL, HT_, 17, GRID) NLC_LIN_CASE(KERNEL, HT_, 21, GRID)        \
  NLC_LIN_CASE(KERNEL, HT_, 25, GRID)
