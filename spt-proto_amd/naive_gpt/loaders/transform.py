"""Text clean-up and fixed-length packing (reference: loaders/transform.py:7-92).

``Sanitize``      whitespace / empty-bracket / punctuation normalisation, paragraph by paragraph
``ClampPadding``  pad to ``seq_length`` or cut a RANDOM window of that length (language modelling)
``TruncPadding``  pad or keep the LAST ``seq_length`` ids and put the unpadded length in front:
                  element 0 of a sample is then the position of its last real token, which is where
                  an MMLU prompt's answer letter sits (script/3-mmlu-evaluate.py:78-91 reads it back)
"""
import random
import re

from torch import nn

_EMPTY_BRACKETS = [re.compile(p) for p in (r'\(\)', r'\[\]', r'\{\}')]
_SPACE_BEFORE_PUNCT = re.compile(r'\s([,.?!;:])')
_SPACES = re.compile(r'\s+')


class Sanitize(nn.Module):
    def forward(self, text: str) -> str:
        assert isinstance(text, str)
        kept = []
        for paragraph in text.split('\n\n'):
            # (the reference re-runs the space clean-up after every rule, transform.py:33-37: an
            # emptied bracket must not leave a space in front of the punctuation behind it)
            for rule in _EMPTY_BRACKETS:
                paragraph = _SPACES.sub(' ', rule.sub(' ', paragraph)).strip()
            paragraph = _SPACES.sub(' ', _SPACE_BEFORE_PUNCT.sub(r'\g<1>', paragraph)).strip()
            if paragraph:
                kept.append(paragraph)
        return '\n\n'.join(kept)


def _padded(sequence, seq_length: int, pad_value):
    return list(sequence) + [pad_value] * (seq_length - len(sequence))


class ClampPadding(nn.Module):
    def __init__(self, seq_length: int, pad_value: int = 0):
        super().__init__()
        self.seq_length, self.pad_value = seq_length, pad_value

    def forward(self, sequence):
        assert isinstance(sequence, (list, tuple))
        extra = len(sequence) - self.seq_length
        if extra <= 0:
            return _padded(sequence, self.seq_length, self.pad_value)
        start = random.randrange(extra + 1)
        return list(sequence[start:start + self.seq_length])


class TruncPadding(nn.Module):
    def __init__(self, seq_length: int, pad_value: int = 0):
        super().__init__()
        self.seq_length, self.pad_value = seq_length, pad_value

    def forward(self, sequence):
        assert isinstance(sequence, (list, tuple))
        tail = list(sequence[-self.seq_length:]) if len(sequence) > self.seq_length else list(sequence)
        return [len(tail)] + _padded(tail, self.seq_length, self.pad_value)
