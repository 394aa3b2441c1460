// One-launch planner body (nlc_fused_kernel.h) for hidden_units = 64, the class default (w_nl.py:72; GRU hidden 32).
#include "nlc_fused_kernel.h"

namespace nlc {
NLC_FUSED_DEFINE_LAUNCHERS(h64, 4, 32, 3, 4)
}  // namespace nlc
