"""Golden vectors for naive_gpt.loaders' text transforms, from the REFERENCE's own
naive_gpt/loaders/transform.py (imported by file path: the package's __init__ needs torchdata /
torchtext / lightning, which this image lacks; transform.py itself needs only re, random, torch).
Run in the build container only (the reference does not travel):

    python tests/golden/make_loaders_golden.py        # -> tests/golden/loaders.json
"""
import importlib.util
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('ref_transform', '/root/reference/naive_gpt/loaders/transform.py')
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

texts = [
    '', '\n\n\n', '  a  b  ', 'x ( ) y', 'one () two [] three {} four',
    'spaces before , punctuation . and ? more ! here ; and : there',
    'para one\nstill one\n\n\n\npara   two  ()  .\n\n   \n\npara three []',
    'tabs\tand\n newlines , mixed ()\n\n()\n\nlast ( ) one !',
    'a (), b [] . c {} ; d',
    'keep (this) and [that] and {those}',
]
out = {'sanitize': [[t, ref.Sanitize()(t)] for t in texts], 'trunc': [], 'clamp': []}
for n, length, pad in [(0, 4, 9), (3, 8, 0), (8, 8, 1), (9, 8, 1), (20, 6, 7)]:
    seq = list(range(100, 100 + n))
    out['trunc'].append([seq, length, pad, ref.TruncPadding(length, pad)(list(seq))])
for seed, n, length, pad in [(0, 3, 8, 255), (1, 8, 8, 0), (2, 30, 8, 0), (3, 9, 8, 5), (4, 64, 16, 0)]:
    seq = list(range(n))
    random.seed(seed)
    out['clamp'].append([seed, seq, length, pad, ref.ClampPadding(length, pad)(list(seq))])
json.dump(out, open(os.path.join(HERE, 'loaders.json'), 'w'), indent=1)
print('wrote loaders.json:', {k: len(v) for k, v in out.items()})
