"""Tokenizers without a network.  The reference fetches `facebook/opt-1.3b` /
`princeton-nlp/Sheared-LLaMA-2.7B` by name (loaders/mmlu.py:28-33); here a tokenizer is an object
with ``encode`` / ``decode`` / ``pad_token_id``, a LOCAL directory for ``transformers`` (never a
download), or ``'bytes'``: a dependency-free byte-level tokenizer for smoke runs and tests."""
import os


class ByteTokenizer:
    """ids 0 .. 255 = the UTF-8 bytes, 256 = pad, 257 = bos (prepended, as the OPT / LLaMA
    tokenizers prepend theirs)."""
    pad_token_id, bos_token_id, vocab_size = 256, 257, 258

    def encode(self, text: str):
        return [self.bos_token_id] + list(text.encode('utf-8'))

    def decode(self, ids) -> str:
        if hasattr(ids, 'tolist'):
            ids = ids.tolist()
        if isinstance(ids, int):
            ids = [ids]
        return bytes(i for i in ids if 0 <= i < 256).decode('utf-8', errors='replace')


def resolve_tokenizer(tokenizer):
    if tokenizer is None or tokenizer == 'bytes':
        return ByteTokenizer()
    if not isinstance(tokenizer, str):
        return tokenizer                                    # already an object
    if not os.path.isdir(tokenizer):
        raise FileNotFoundError(
            'tokenizer {!r}: give a local directory (a saved transformers tokenizer), an object, or '
            "'bytes' -- nothing is downloaded (the reference names 'facebook/opt-1.3b' / "
            "'princeton-nlp/Sheared-LLaMA-2.7B', loaders/mmlu.py:28-33)".format(tokenizer))
    import transformers
    return transformers.AutoTokenizer.from_pretrained(tokenizer, local_files_only=True)
