"""Flan-Mini (one JSON document per line; reference: loaders/details/flanmini.py:7-45,
loaders/flanmini.py) and the weighted mix of endless streams the MMLU train loader uses
(loaders/details/concat.py:5-27)."""
import json
import random

from torch import nn
from torch.utils import data

from .mmlu import _DataModule, _Encode
from .reader import LineReader
from .tokenizer import resolve_tokenizer
from .transform import TruncPadding


class _JsonThen(nn.Module):
    def __init__(self, then=None):
        super().__init__()
        self.then = then

    def forward(self, line: str):
        doc = json.loads(line)
        return doc if self.then is None else self.then(doc)


class FlanMiniDataset(LineReader):
    FILES = {'flan-mini/flan_mini.jsonl': 1.0}

    def __init__(self, root: str, mode: str, shuffle: bool = True, min_length: int = 64,
                 buffer_size: int = 16384, text_transform=None):
        if mode not in ('train', 'valid'):
            raise RuntimeError('mode: train | valid')
        super().__init__(root=root, files=dict(self.FILES), shuffle=shuffle, min_length=min_length,
                         buffer_size=buffer_size, return_path=False,
                         text_transform=_JsonThen(text_transform))


class WeightedMix(data.IterableDataset):
    """{dataset: weight}: every draw takes the next item of one source, chosen in proportion to its
    weight; sources restart when they end (``infinite``), the mix passes through a small reservoir."""

    def __init__(self, datasets: dict, buffer_size: int = 1024, infinite: bool = True):
        super().__init__()
        self.datasets, self.buffer_size, self.infinite = datasets, buffer_size, infinite

    def __iter__(self):
        rng = random.Random()
        sources = {d: iter(d) for d in self.datasets}
        buffer = []
        while sources:
            keys = list(sources)
            pick = rng.choices(keys, weights=[self.datasets[d] for d in keys])[0]
            try:
                item = next(sources[pick])
            except StopIteration:
                if self.infinite:
                    sources[pick] = iter(pick)
                    continue
                del sources[pick]
                continue
            if len(buffer) < self.buffer_size:
                buffer.append(item)
                continue
            slot = rng.randrange(self.buffer_size)
            buffer[slot], item = item, buffer[slot]
            yield item
        rng.shuffle(buffer)
        yield from buffer


class FlanMiniDataModule(_DataModule):
    def __init__(self, root: str, seq_length: int, batch_size: int, num_workers: int = 0, tokenizer='bytes'):
        self.root, self.seq_length = root, seq_length
        self.batch_size, self.num_workers = batch_size, num_workers
        self.tokenizer = resolve_tokenizer(tokenizer)
        self.pad_value = getattr(self.tokenizer, 'pad_token_id', None) or 0

    def _dataset(self, mode: str):
        encode = _Encode(self.tokenizer, TruncPadding(self.seq_length, self.pad_value))
        return FlanMiniDataset(self.root, mode='valid' if mode == 'test' else mode, text_transform=encode)
