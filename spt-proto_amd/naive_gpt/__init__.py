"""naive_gpt -- MI355X-native drop-in for the hot path of ytgui/SPT-proto.

Same import surface as the reference package for the PQ sparse-attention /
routed-FFN path (``ext``, ``kernels``, ``layers``, ``utils``) plus the two decoder
models and the checkpoint format the fine-tuning recipe consumes (``models``,
SURVEY.md 8 f-2) and its data pipeline over LOCAL paths (``loaders``, SURVEY.md 8 f-4: nothing is
downloaded; imported on first use -- it pulls in nothing the hot path needs).
"""
from . import ext
from . import kernels
from . import layers
from . import utils
from . import models

__all__ = ['ext', 'kernels', 'layers', 'utils', 'models', 'loaders']


def __getattr__(name):
    if name == 'loaders':
        import importlib
        return importlib.import_module('.loaders', __name__)
    raise AttributeError(name)
