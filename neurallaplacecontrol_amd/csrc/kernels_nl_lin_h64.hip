// LIN instances of the rollout kernels (kernels_nl_lin.hip), hidden width 64.
#include "nlc_nl_lin_launch.h"

namespace nlc {

NLC_DEFINE_LIN_ROLLOUT_LAUNCHER(h64, 4)

}  // namespace nlc
