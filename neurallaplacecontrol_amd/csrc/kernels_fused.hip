// One-launch planner body (nlc_fused_kernel.h), hidden width 128 (the harness's hidden_units, train_utils.py:29-54), and the
// width dispatch of the launcher.
#include "nlc_fused_kernel.h"

namespace nlc {

NLC_FUSED_DEFINE_LAUNCHERS(h128, 8, 64, 3, 4)
hipError_t fused_max_resident_blocks_h64(int bpc_built, int* blocks_per_cu);
hipError_t launch_nl_plan_fused_h64(const FusedArgs& a, unsigned grid, int bpc_built, hipStream_t s);
hipError_t fused_max_resident_blocks_h256(int bpc_built, int* blocks_per_cu);
hipError_t launch_nl_plan_fused_h256(const FusedArgs& a, unsigned grid, int bpc_built, hipStream_t s);

hipError_t fused_max_resident_blocks(int h, int bpc_built, int* blocks_per_cu) {
  switch (h) {
    case 64: return fused_max_resident_blocks_h64(bpc_built, blocks_per_cu);
    case 128: return fused_max_resident_blocks_h128(bpc_built, blocks_per_cu);
    case 256: return fused_max_resident_blocks_h256(bpc_built, blocks_per_cu);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_nl_plan_fused(const FusedArgs& a, int g, unsigned grid, int bpc_built, hipStream_t s) {
  if (a.r.K <= 0) return hipSuccess;
  if (2 * g != a.r.net.h) return hipErrorInvalidValue;
  switch (a.r.net.h) {
    case 64: return launch_nl_plan_fused_h64(a, grid, bpc_built, s);
    case 128: return launch_nl_plan_fused_h128(a, grid, bpc_built, s);
    case 256: return launch_nl_plan_fused_h256(a, grid, bpc_built, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace nlc

// tools/split_phase_clocks.py (a -DNLC_PHASE_CLOCKS=1 build of this unit): phase sums of the latency-split bodies launched from here
namespace nlc {
NLC_DEFINE_SPLIT_CLK_READER(nlc_debug_split_clocks_fused)
}
