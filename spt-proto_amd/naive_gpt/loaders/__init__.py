"""Local-path data pipeline of the SPT recipe (SURVEY f-4): the names of the reference's
``naive_gpt.loaders`` (loaders/__init__.py:1-16) without torchdata / torchtext / Lightning, none
of which this image has, and without any download: datasets are directories the caller names,
tokenizers are objects or LOCAL paths (``transformers`` is asked with ``local_files_only``)."""
# basic
from .transform import Sanitize, ClampPadding, TruncPadding
# readers
from .reader import LineReader, TextFolder
# datasets and their loaders
from .mmlu import MMLUPrompt, MMLUDataset, MMLUDataModule
from .flanmini import FlanMiniDataset, FlanMiniDataModule
from .wikitext import WikitextDataModule
from .tokenizer import ByteTokenizer, resolve_tokenizer

__all__ = ['Sanitize', 'ClampPadding', 'TruncPadding', 'LineReader', 'TextFolder', 'MMLUPrompt',
           'MMLUDataset', 'MMLUDataModule', 'FlanMiniDataset', 'FlanMiniDataModule',
           'WikitextDataModule', 'ByteTokenizer', 'resolve_tokenizer']
