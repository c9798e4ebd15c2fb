"""naive_gpt -- MI355X-native drop-in for the hot path of ytgui/SPT-proto.

Same import surface as the reference package for the PQ sparse-attention /
routed-FFN path (``ext``, ``kernels``, ``layers``, ``utils``); the reference's
loaders and model zoo are out of scope.
"""
from . import ext
from . import kernels
from . import layers
from . import utils

__all__ = ['ext', 'kernels', 'layers', 'utils']
