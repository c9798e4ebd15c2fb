"""naive_gpt -- MI355X-native drop-in for the hot path of ytgui/SPT-proto.

Same import surface as the reference package for the PQ sparse-attention /
routed-FFN path (``ext``, ``kernels``, ``layers``, ``utils``) plus the two decoder
models and the checkpoint format the fine-tuning recipe consumes (``models``,
SURVEY.md 8 f-2).  The reference's dataset loaders are out of scope.
"""
from . import ext
from . import kernels
from . import layers
from . import utils
from . import models

__all__ = ['ext', 'kernels', 'layers', 'utils', 'models']
