// LIN instances of the rollout kernels, hidden width 128 (the harness's hidden_units; kernels_nl_lin_h64.hip / _h256.hip: the
// other widths): fixed Talbot / Stehfest models on the fused rollout (round 3).  Their reconstruction x = (1/t) sum_k (w_re,k Re F_k - w_im,k Im F_k) is linear in F like the Fourier
// sum, so it rides the MFMA epilogue too -- with BOTH components of F = R e^{i theta} per term: two ILT MFMAs per slot group
// against the coefficient fragments NlNetArgs::Cp (w_re / t) and Cp2 (-w_im / t), which nlc_mppi_configure folds for the
// planner's constant prediction time.  A translation unit of its own: the Fourier instances keep their code and registers.
#include "nlc_nl_lin_launch.h"

namespace nlc {

NLC_DEFINE_LIN_ROLLOUT_LAUNCHER(h128, 8)

}  // namespace nlc
