"""Weighted, endless, shuffled text streams over local files (reference: loaders/reader.py:9-148,
built there on torchdata datapipes: FileOpener -> CSVParser | LineReader -> Cycler ->
SampleMultiplexer -> Mapper(clean) -> Filter(min_length) -> Shuffler).  Plain generators here."""
import csv
import os
import random

import torch
from torch.utils import data

from .transform import Sanitize


def _records(path: str, reader: str, skip_lines: int):
    """One pass over a file: lines (without their newline) or csv rows (lists of fields)."""
    with open(path, 'r', encoding='utf-8', newline='' if reader == 'csv' else None) as f:
        rows = csv.reader(f) if reader == 'csv' else (line.rstrip('\n') for line in f)
        for i, row in enumerate(rows):
            if i >= skip_lines:
                yield row


def _forever(path: str, reader: str, skip_lines: int):
    while True:
        empty = True
        for row in _records(path, reader, skip_lines):
            empty = False
            yield path, row
        if empty:
            return


class LineReader(data.IterableDataset):
    """``files``: {path relative to root: sampling weight}.  Every file is cycled for ever; each
    draw picks a file with probability proportional to its weight (a file that runs dry -- an
    empty one -- drops out); rows are sanitised, rows shorter than ``min_length`` characters are
    dropped; ``shuffle`` passes the stream through a reservoir of ``buffer_size`` items."""

    def __init__(self, root: str, files: dict, reader: str = 'line', shuffle: bool = True,
                 skip_lines: int = 0, min_length: int = 64, buffer_size: int = 16384,
                 return_path: bool = False, append_path: bool = False,
                 text_transform=None, path_transform=None):
        super().__init__()
        if reader not in ('line', 'csv'):
            raise RuntimeError('reader: line | csv')
        self.files = {os.path.join(root.rstrip('/'), p.lstrip('/')): w for p, w in files.items()}
        self.reader, self.shuffle, self.skip_lines = reader, shuffle, skip_lines
        self.min_length, self.buffer_size = min_length, buffer_size
        self.return_path, self.append_path = return_path, append_path
        self.text_transform, self.path_transform = text_transform, path_transform
        self._clean = Sanitize()

    # -- the stream before the transforms: (path, cleaned content) --
    def _multiplexed(self, rng: random.Random):
        streams = {p: _forever(p, self.reader, self.skip_lines) for p in self.files}
        weights = dict(self.files)
        while streams:
            paths = list(streams)
            path = rng.choices(paths, weights=[weights[p] for p in paths])[0]
            try:
                yield next(streams[path])
            except StopIteration:
                del streams[path]

    def _cleaned(self, rng: random.Random):
        for path, content in self._multiplexed(rng):
            if isinstance(content, str):
                content = self._clean(content)
                size = len(content)
            else:
                content = [self._clean(field) for field in content]
                size = sum(len(field) for field in content)
            if size >= self.min_length:
                yield path, content

    def _shuffled(self, rng: random.Random):
        source = self._cleaned(rng)
        if not self.shuffle:
            yield from source
            return
        buffer = []
        for item in source:
            if len(buffer) < self.buffer_size:
                buffer.append(item)
                continue
            slot = rng.randrange(self.buffer_size)
            buffer[slot], item = item, buffer[slot]
            yield item
        rng.shuffle(buffer)
        yield from buffer

    def __iter__(self):
        # a fresh seed per iterator, also handed to `random` / torch (the transforms draw from
        # them: ClampPadding's window), as the reference does (reader.py:104-110)
        seed = int.from_bytes(os.urandom(4), byteorder='little')
        random.seed(seed)
        torch.random.manual_seed(seed)
        for path, content in self._shuffled(random.Random(seed)):
            item = (content, path) if self.append_path else content
            if self.text_transform is not None:
                item = self.text_transform(item)
            if self.path_transform is not None:
                path = self.path_transform(path)
            yield (item, path) if self.return_path else item


class TextFolder(LineReader):
    """Every file under ``root``, weighted by its size in bytes (reference: reader.py:127-148)."""

    def __init__(self, root: str, reader: str = 'line', shuffle: bool = True, skip_lines: int = 0,
                 min_length: int = 64, buffer_size: int = 16384, return_path: bool = True,
                 append_path: bool = False, text_transform=None, path_transform=None):
        root = root.rstrip('/')
        files = {}
        for folder, _, names in sorted(os.walk(root)):
            for name in sorted(names):
                full = os.path.join(folder, name)
                files[full[len(root):]] = os.stat(full).st_size
        if not files:
            raise FileNotFoundError('no files under {!r}'.format(root))
        super().__init__(root=root, files=files, reader=reader, shuffle=shuffle,
                         skip_lines=skip_lines, min_length=min_length, buffer_size=buffer_size,
                         return_path=return_path, append_path=append_path,
                         text_transform=text_transform, path_transform=path_transform)
