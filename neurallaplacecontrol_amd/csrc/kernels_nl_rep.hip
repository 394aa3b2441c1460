// The representation function alone (nl_repfunc_kernel, nl_repfunc_split_kernel) for hidden_units = 128 (see kernels_nl.hip).
#include "nlc_nl_kernels.h"

namespace nlc {

hipError_t launch_nl_repfunc_h128(const RepFuncArgs& a, hipStream_t s) {
  if (!a.general_t && a.slot_major && !a.write_angles && a.split) {
    const unsigned g16 = (unsigned)((a.N + 15) / 16);
    switch (a.net.nt3) {
#define X(N)                                                                                  \
  case N:                                                                                     \
    hipLaunchKernelGGL((nl_repfunc_split_kernel<8, N>), dim3(g16), dim3(256), 0, s, a);   \
    break;
      NLC_FOR_NT3(X)
#undef X
      default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  const unsigned grid = (unsigned)((a.N + 63) / 64);
  switch (a.net.nt3) {
#define X(N)                                                                                          \
  case N:                                                                                             \
    if (a.general_t) {                                                                                \
      hipLaunchKernelGGL((nl_repfunc_kernel<8, N, true>), dim3(grid), dim3(256), 0, s, a);        \
    } else {                                                                                          \
      hipLaunchKernelGGL((nl_repfunc_kernel<8, N, false>), dim3(grid), dim3(256), 0, s, a);       \
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

// tools/split_phase_clocks.py (a -DNLC_PHASE_CLOCKS=1 build of this unit): phase sums of the latency-split bodies launched from here
namespace nlc {
NLC_DEFINE_SPLIT_CLK_READER(nlc_debug_split_clocks_rep)
}
