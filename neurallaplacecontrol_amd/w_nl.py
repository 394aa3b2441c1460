"""Import-path twin of the reference's ``w_nl.py``: ``from w_nl import NeuralLaplaceModel`` keeps working
when this package directory is put on ``sys.path``.  Implementation: nl_model.py."""

from .nl_model import (  # noqa: F401
    LaplaceRepresentationFunc,
    NeuralLaplaceModel,
    ReverseGRUEncoder,
    load_replay_buffer,
)
