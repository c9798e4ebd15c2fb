"""MMLU as the recipe reads it (reference: loaders/details/mmlu.py:79-170, loaders/mmlu.py:12-79).

Directory layout under ``root`` (the hendrycks/test archive, unpacked): ``mmlu/dev/*.csv`` (the
few-shot pool), ``mmlu/val``, ``mmlu/test``, ``mmlu/auxiliary_train``; a csv row is
``question, A, B, C, D, answer``.  A sample is ``n_shots`` prompts drawn from the pool, then the
item's own prompt, joined by blank lines and ENDING in the answer letter:

    The following are multiple choice questions (with answers) about <subject>
    <question>
    A. ..      B. ..      C. ..      D. ..
    Answer: <letter>

Tokenised and packed with ``TruncPadding`` the sample is ``[n, id_1 .. id_n, pad ..]``: ``batch[:, 0]``
is the position of the answer letter, which is how script/3-mmlu-evaluate.py:78-91 (and
``utils.SparseTuner.validation_step``) score it."""
import os

import torch
from torch import nn
from torch.utils import data

from .reader import TextFolder
from .tokenizer import resolve_tokenizer
from .transform import TruncPadding

SPLITS = {'test': 'test', 'valid': 'val', 'train': 'auxiliary_train'}
HEADER = 'The following are multiple choice questions (with answers) about'


class MMLUPrompt(nn.Module):
    prompt = HEADER

    def forward(self, item):
        row, path = item
        if len(row) != 6:
            raise RuntimeError('an MMLU row has six fields (question, four choices, answer): {!r}'.format(row))
        # "high_school_physics_test.csv" -> "high school physics"
        subject = ' '.join(os.path.basename(path).split('_')[:-1])
        lines = [row[0]] + ['{}. {}'.format(letter, choice) for letter, choice in zip('ABCD', row[1:5])]
        return '{} {}\n{}\nAnswer: {}'.format(self.prompt, subject, '\n'.join(lines), row[5])


class MMLUDataset(data.IterableDataset):
    def __init__(self, root: str, mode: str, n_shots: int = 0, shuffle: bool = True,
                 min_length: int = 64, buffer_size: int = 16384, return_path: bool = False,
                 text_transform=None, path_transform=None):
        super().__init__()
        if mode not in SPLITS:
            raise RuntimeError('mode: test | valid | train')
        self.n_shots, self.return_path = n_shots, return_path
        self.text_transform, self.path_transform = text_transform, path_transform
        base = os.path.join(root.rstrip('/'), 'mmlu')
        common = dict(reader='csv', skip_lines=0, min_length=min_length, buffer_size=buffer_size,
                      append_path=True, text_transform=MMLUPrompt())
        self.context = TextFolder(root=os.path.join(base, 'dev'), shuffle=True, return_path=False, **common)
        self.dataset = TextFolder(root=os.path.join(base, SPLITS[mode]), shuffle=shuffle,
                                  return_path=return_path, **common)

    def __iter__(self):
        shots = iter(self.context)
        for item in self.dataset:
            content, filename = item if self.return_path else (item, None)
            content = '\n\n'.join([next(shots) for _ in range(self.n_shots)] + [content])
            if self.text_transform is not None:          # (after the few-shot prompts are in place)
                content = self.text_transform(content)
            if self.path_transform is not None:
                filename = self.path_transform(filename)
            yield (content, filename) if self.return_path else content


class _Encode(nn.Module):
    """text -> [n, ids .., pad ..] as an int64 tensor"""

    def __init__(self, tokenizer, packer):
        super().__init__()
        self.tokenizer, self.packer = tokenizer, packer

    def forward(self, text: str):
        return torch.tensor(self.packer(list(self.tokenizer.encode(text))), dtype=torch.long)


class _DataModule:
    """What the scripts use of a LightningDataModule: three loader factories."""
    batch_size: int
    num_workers: int

    def _dataset(self, mode: str):
        raise NotImplementedError

    def _dataloader(self, mode: str):
        return data.DataLoader(self._dataset(mode), shuffle=False, batch_size=self.batch_size,
                               num_workers=self.num_workers, pin_memory=torch.cuda.is_available())

    def train_dataloader(self):
        return self._dataloader('train')

    def val_dataloader(self):
        return self._dataloader('valid')

    def test_dataloader(self):
        return self._dataloader('test')


class MMLUDataModule(_DataModule):
    """``tokenizer``: an object, a local directory, or 'bytes' (loaders/tokenizer.py).  'train' mixes
    the auxiliary MMLU split (weight 0.1) with Flan-Mini (weight 1.0) as the reference does
    (loaders/mmlu.py:49-62) when ``flan-mini/flan_mini.jsonl`` exists under ``root``."""

    def __init__(self, root: str, n_shots: int, seq_length: int, batch_size: int, num_workers: int = 0,
                 tokenizer='bytes'):
        self.root, self.n_shots, self.seq_length = root, n_shots, seq_length
        self.batch_size, self.num_workers = batch_size, num_workers
        self.tokenizer = resolve_tokenizer(tokenizer)
        self.pad_value = getattr(self.tokenizer, 'pad_token_id', None) or 0

    def _dataset(self, mode: str):
        encode = _Encode(self.tokenizer, TruncPadding(self.seq_length, self.pad_value))
        mmlu = MMLUDataset(self.root, mode=mode, n_shots=self.n_shots, text_transform=encode)
        flan = os.path.join(self.root.rstrip('/'), 'flan-mini', 'flan_mini.jsonl')
        if mode != 'train' or not os.path.isfile(flan):
            return mmlu
        from .flanmini import FlanMiniDataset, WeightedMix
        return WeightedMix({mmlu: 0.1, FlanMiniDataset(self.root, mode=mode, text_transform=encode): 1.0})
