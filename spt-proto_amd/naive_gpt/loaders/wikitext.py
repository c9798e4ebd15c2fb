"""WikiText-103 raw files as random fixed-length windows (reference: loaders/wikitext.py:9-73)."""
from .mmlu import _DataModule, _Encode
from .reader import LineReader
from .tokenizer import resolve_tokenizer
from .transform import ClampPadding

FILES = {'test': 'wikitext-103/wiki.test.raw', 'train': 'wikitext-103/wiki.train.raw',
         'valid': 'wikitext-103/wiki.valid.raw'}


class WikitextDataModule(_DataModule):
    def __init__(self, root: str, seq_length: int, batch_size: int, num_workers: int = 0, tokenizer='bytes'):
        self.root, self.seq_length = root, seq_length
        self.batch_size, self.num_workers = batch_size, num_workers
        self.tokenizer = resolve_tokenizer(tokenizer)
        self.pad_value = getattr(self.tokenizer, 'pad_token_id', None) or 0

    def _dataset(self, mode: str):
        encode = _Encode(self.tokenizer, ClampPadding(self.seq_length, self.pad_value))
        return LineReader(root=self.root, files={FILES[mode]: 1.0}, shuffle=True, text_transform=encode)

    def predict_dataloader(self):
        return self._dataloader('test')
