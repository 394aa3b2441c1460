// Representation function alone (nl_repfunc_kernel) for hidden_units = 256 (see kernels_nl.hip; a translation unit of its own: the width-256 instances are the
// longest compiles of the library, and the build is as long as its longest unit).
#include "nlc_nl_kernels.h"

namespace nlc {

hipError_t launch_nl_repfunc_h256(const RepFuncArgs& a, hipStream_t s) {
  const unsigned grid = (unsigned)((a.N + 63) / 64);
  switch (a.net.nt3) {
#define X(N)                                                                                          \
  case N:                                                                                             \
    if (a.general_t) {                                                                                \
      hipLaunchKernelGGL((nl_repfunc_kernel<16, N, true>), dim3(grid), dim3(256), 0, s, a);        \
    } else {                                                                                          \
      hipLaunchKernelGGL((nl_repfunc_kernel<16, N, false>), dim3(grid), dim3(256), 0, s, a);       \
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
