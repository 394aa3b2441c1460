// Single model forward (nl_forward_kernel) for hidden_units = 256, one shared query time (see kernels_nl.hip; a
// translation unit of its own: the build is as long as its longest unit).
#include "nlc_nl_kernels.h"

namespace nlc {

hipError_t launch_nl_forward_h256_const(const ForwardArgs& a, hipStream_t s) {
  const unsigned grid = (unsigned)((a.N + 63) / 64);
  switch (a.net.nt3) {
#define X(N)                                                                                            \
  case N:                                                                                               \
    hipLaunchKernelGGL((nl_forward_kernel<16, N, false>), dim3(grid), dim3(256), 0, s, a); \
    break;
    NLC_FOR_NT3(X)
#undef X
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace nlc
