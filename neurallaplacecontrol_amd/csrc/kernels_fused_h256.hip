// One-launch planner body (nlc_fused_kernel.h) for hidden_units = 256 (GRU hidden 128): 64 KB of cooperative hidden-state
// images / 68 KB of rollout exchange per workgroup -> two workgroups per CU (256 VGPRs), the fused body's minimum.
#include "nlc_fused_kernel.h"

namespace nlc {
NLC_FUSED_DEFINE_LAUNCHERS(h256, 16, 128, 2, 2)
}  // namespace nlc
