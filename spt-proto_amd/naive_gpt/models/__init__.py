"""naive_gpt.models -- the two decoder-only language models of the reference
(``naive_gpt/models/__init__.py:1-5``)."""
from .decoder import DecoderLM
from .opt import OPTModel
from .llama import LLaMAModel

__all__ = ['DecoderLM', 'OPTModel', 'LLaMAModel']
