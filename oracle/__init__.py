"""CPU oracle for the Neural-Laplace-Control planning hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a float64 torch-CPU restatement of the reference algorithm
(samholt/NeuralLaplaceControl: ``planners/mppi_delay.py``, ``w_nl.py``,
``oracle.py``, the env reward functions, plus the external ``torchlaplace``
ILT).  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and there only as the checker / the timed
CPU baseline.  Nothing under ``neurallaplacecontrol_amd/`` imports it.

Parity status
-------------
* MPPI (a1-a4, a11), oracle dynamics, env costs, GRU encoder, representation
  MLP, model plumbing (a5-a8, a10, a12): PINNED against the imported reference
  classes -- see ``tests/golden/make_golden.py`` and the committed fixtures.
* Delta-t RNN baseline (``rnn_model.py``): PINNED against the reference ``DeltaTRNN``
  class (``tests/golden/make_golden_rnn.py``, fixtures ``g9_dtrnn_*``).
* NODE baseline (``node_model.py``): MLP, normalisation, augmentation and planner PINNED against the reference
  classes (G11); its ``torchdiffeq.odeint`` call is restated (fixed-grid Euler): **parity unpinned for odeint**.
* ``laplace_reconstruct`` body (a9): **parity unpinned vs upstream
  torchlaplace** -- the PyPI package ``torchlaplace`` (unpinned in the
  reference's ``requirements.txt:17``) is absent from the build container and
  cannot be fetched.  The restatement follows the Neural Laplace paper
  (arXiv 2206.04843) and de Hoog-Knight-Stokes 1982 as coded in mpmath 1.3.0
  ``calculus/inverselaplace.py:356-537``; it is pinned algorithmically against
  analytic Laplace pairs and ``mpmath.invertlaplace(method='dehoog')``.
"""

from . import envs, ilt, mppi, nl_model, node_model, rnn_model  # noqa: F401
