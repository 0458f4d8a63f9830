"""alignn_amd: MI355X (gfx950) native implementation of ALIGNN's message-passing hot path.

Public surface mirrors ``alignn.models.alignn`` of the reference; see ``alignn_amd.alignn``.
"""

from .alignn import ALIGNN, ALIGNNConfig, ALIGNNConv, EdgeGatedGraphConv, MLPLayer, RBFExpansion  # noqa: F401
from .alignn_atomwise import ALIGNNAtomWise, ALIGNNAtomWiseConfig  # noqa: F401
from .ealignn_atomwise import eALIGNNAtomWise, eALIGNNAtomWiseConfig  # noqa: F401
from .graph import CSRGraph, GraphBatch, build_csr  # noqa: F401

__version__ = "0.1.0"
