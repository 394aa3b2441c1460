"""neurallaplacecontrol_amd -- the Neural-Laplace-Control planning hot path on AMD MI355X (gfx950).

Drop-in mirrors of the reference's Python interfaces for this path:

    from neurallaplacecontrol_amd import MPPIDelay              # planners/mppi_delay.py:54
    from neurallaplacecontrol_amd import NeuralLaplaceModel     # w_nl.py:66
    from neurallaplacecontrol_amd import laplace_reconstruct    # torchlaplace (external)
    from neurallaplacecontrol_amd import DeltaTRNN, NODE        # train_utils.py:589, :664 (baseline models)

All arithmetic runs in hand-written HIP kernels behind the C ABI of ``libnlc_hip.so``
(``include/nlc.h``); importing the package does not touch the GPU.  There is no CPU fallback.
"""

from ._lib import set_default_options  # noqa: F401
from .envs import EnvCost, NLDynamics, OracleDynamics, initial_state, noise_sigma  # noqa: F401
from .env_loop import BatchedEnv  # noqa: F401
from .laplace import ilt_reconstruct, laplace_reconstruct, rep_func_inputs  # noqa: F401
from .nl_model import LaplaceRepresentationFunc, NeuralLaplaceModel, ReverseGRUEncoder  # noqa: F401
from .node_model import NODE, xOdeFuncInXAndU  # noqa: F401
from .rnn_model import RNN, DeltaTRNN  # noqa: F401
from .planners.mppi_batch import BatchedMPPIDelay  # noqa: F401
from .planners.mppi_delay import MPPIDelay  # noqa: F401

__all__ = [
    "MPPIDelay",
    "BatchedMPPIDelay",
    "BatchedEnv",
    "NeuralLaplaceModel",
    "DeltaTRNN",
    "RNN",
    "NODE",
    "ReverseGRUEncoder",
    "LaplaceRepresentationFunc",
    "laplace_reconstruct",
    "ilt_reconstruct",
    "rep_func_inputs",
    "NLDynamics",
    "OracleDynamics",
    "EnvCost",
    "noise_sigma",
    "initial_state",
    "set_default_options",
]
