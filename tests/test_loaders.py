"""naive_gpt.loaders / utils.evaluate (SURVEY f-4): the reference's data pipeline over LOCAL paths.

Pinned by: tests/golden/loaders.json (the reference's own transform.py on sample inputs, made by
tests/golden/make_loaders_golden.py), the known answer of the reference's test/loader/
test_plaintext.py:9-33, the sample invariant of test/loader/test_mmlu.py:24-40 (element 0 points at
a decoded 'A'..'D'), and the prompt format of loaders/details/mmlu.py:79-106 read from the source."""
import csv
import json
import os
import random

import torch

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'loaders.json')))


def test_sanitize_and_paddings_equal_the_reference_on_its_vectors():
    from naive_gpt import loaders
    for text, want in GOLDEN['sanitize']:
        assert loaders.Sanitize()(text) == want, repr(text)
    for seq, length, pad, want in GOLDEN['trunc']:
        assert loaders.TruncPadding(length, pad)(list(seq)) == want
    for seed, seq, length, pad, want in GOLDEN['clamp']:
        random.seed(seed)                      # (the window is drawn from `random`, as in the reference)
        assert loaders.ClampPadding(length, pad)(list(seq)) == want


def test_sanitize_known_answer_of_the_reference_test():
    """test/loader/test_plaintext.py:9-33."""
    from naive_gpt import loaders
    for i in range(10):
        assert loaders.Sanitize()('\n' * i) == ''
    src = """
      [] Advanced Micro Devices, Inc., commonly abbreviated as AMD, is an American multinational
    semiconductor company based in Santa  Clara , California (), that develops computer processors
    and related technologies for business and consumer markets.\n\n\n\n\n
    AMD's main products include microprocessors, motherboard chipsets, embedded processors, graphics
    processors, and FPGAs for servers, workstations, personal computers, and embedded system applications.\n
    """
    want = ('Advanced Micro Devices, Inc., commonly abbreviated as AMD, is an American multinational '
            'semiconductor company based in Santa Clara, California, that develops computer processors and '
            'related technologies for business and consumer markets.\n\n'
            "AMD's main products include microprocessors, motherboard chipsets, embedded processors, graphics "
            'processors, and FPGAs for servers, workstations, personal computers, and embedded system applications.')
    assert loaders.Sanitize()(src) == want
    y = loaders.ClampPadding(seq_length=40, pad_value=0xff)([random.randint(0, 100) for _ in range(20)])
    assert len(y) == 40 and y[-1] == 0xff


def _write_mmlu(root, n_dev=6, n_rows=12):
    """A tiny MMLU tree: two subjects, rows long enough to pass the 64-character filter."""
    rng = random.Random(0)
    truth = {}
    for split, suffix in (('dev', 'dev'), ('val', 'val'), ('test', 'test'), ('auxiliary_train', 'train')):
        os.makedirs(os.path.join(root, 'mmlu', split), exist_ok=True)
        for subject in ('high_school_physics', 'world_religions'):
            path = os.path.join(root, 'mmlu', split, '{}_{}.csv'.format(subject, suffix))
            with open(path, 'w', newline='') as f:
                w = csv.writer(f)
                for i in range(n_dev if split == 'dev' else n_rows):
                    q = 'Question {} of {} in {}: which of these statements, if any, holds ( ) ?'.format(i, subject, split)
                    choices = ['choice {} number {}'.format(c, rng.randrange(1000)) for c in 'wxyz']
                    answer = rng.choice('ABCD')
                    w.writerow([q] + choices + [answer])
                    truth[q.replace(' ?', '?')] = answer              # (as Sanitize leaves the question)
    return truth


def test_mmlu_prompt_format():
    from naive_gpt import loaders
    row = ['What is 1 + 1?', '1', '2', '3', '4', 'B']
    text = loaders.MMLUPrompt()((row, '/data/mmlu/test/elementary_mathematics_test.csv'))
    assert text == ('The following are multiple choice questions (with answers) about elementary mathematics\n'
                    'What is 1 + 1?\nA. 1\nB. 2\nC. 3\nD. 4\nAnswer: B')
    try:
        loaders.MMLUPrompt()((row[:5], 'x_test.csv'))
        assert False
    except RuntimeError:
        pass


def test_mmlu_samples_point_at_their_answer_letter(tmp_path):
    """The invariant of test/loader/test_mmlu.py:24-40 -- sample[0] is the position of the answer
    letter -- on a local tree with the byte tokenizer, plus the few-shot structure."""
    from naive_gpt import loaders
    truth = _write_mmlu(str(tmp_path))
    seq_length, n_shots = 1536, 3
    dm = loaders.MMLUDataModule(root=str(tmp_path), n_shots=n_shots, seq_length=seq_length + 1, batch_size=1,
                                num_workers=0, tokenizer='bytes')
    seen = 0
    for batch in dm.val_dataloader():
        assert batch.shape == (1, seq_length + 2) and batch.dtype == torch.long
        sample = batch[0]
        pos = int(sample[0])
        letter = dm.tokenizer.decode(sample[pos])
        assert letter in 'ABCD'
        text = dm.tokenizer.decode(sample[1:pos + 1])
        prompts = text.split('\n\n')
        assert len(prompts) == n_shots + 1
        assert all(p.startswith(loaders.MMLUPrompt.prompt) for p in prompts)
        assert all(' in dev:' in p for p in prompts[:-1]) and ' in val:' in prompts[-1]
        question = prompts[-1].split('\n')[1]
        assert truth[question] == letter                      # the item's own answer, not a shot's
        assert (sample[pos + 1:] == dm.pad_value).all()
        seen += 1
        if seen == 16:
            break
    assert seen == 16          # the stream is endless (files are cycled)


def test_trunc_keeps_the_tail_so_the_answer_survives(tmp_path):
    from naive_gpt import loaders
    _write_mmlu(str(tmp_path))
    dm = loaders.MMLUDataModule(root=str(tmp_path), n_shots=5, seq_length=129, batch_size=2, tokenizer='bytes')
    batch = next(iter(dm.test_dataloader()))
    assert batch.shape == (2, 130)
    for sample in batch:
        assert int(sample[0]) == 129 and dm.tokenizer.decode(sample[129]) in 'ABCD'


def test_line_reader_weights_filter_and_cycle(tmp_path):
    from naive_gpt import loaders
    for name, n in (('a.txt', 3), ('b.txt', 3)):
        with open(tmp_path / name, 'w') as f:
            for i in range(n):
                f.write('{} line {} {}\n'.format(name, i, 'x' * 70))
            f.write('short\n')
    reader = loaders.LineReader(str(tmp_path), {'a.txt': 9.0, 'b.txt': 1.0}, shuffle=True, buffer_size=8,
                                return_path=True)
    counts = {'a.txt': 0, 'b.txt': 0}
    for i, (text, path) in enumerate(reader):
        assert len(text) >= 64 and 'short' not in text
        counts[os.path.basename(path)] += 1
        if i == 999:
            break
    assert 800 < counts['a.txt'] < 980 and counts['b.txt'] > 20       # 9 : 1, endlessly
    folder = loaders.TextFolder(str(tmp_path), shuffle=False, return_path=False)
    first = [next(iter(folder)) for _ in range(2)]
    assert all(isinstance(t, str) for t in first)


def test_evaluate_mmlu_reproduces_the_reference_metrics(tmp_path):
    """utils.evaluate_mmlu = script/3-mmlu-evaluate.py:64-93 on a dense model (CPU): perplexity of
    batch[:, 2:] under the logits of batch[:, 1:-1], accuracy with the reference's [B, B] indexing."""
    from naive_gpt import loaders, models, utils
    _write_mmlu(str(tmp_path))
    torch.manual_seed(0)
    tok = loaders.ByteTokenizer()
    model = models.OPTModel(d_model=32, n_heads=2, n_layers=1, max_length=96, vocab_size=tok.vocab_size,
                            d_feedforward=64, p_dropout=0.0)
    dm = loaders.MMLUDataModule(root=str(tmp_path), n_shots=0, seq_length=65, batch_size=2, tokenizer=tok)
    batches = [b for _, b in zip(range(3), dm.test_dataloader())]
    got = utils.evaluate_mmlu(model, batches, n_batches=3)
    assert got['batches'] == 3 and model.training             # the mode is handed back
    loss = ppl = acc = 0.0
    model.eval()
    with torch.no_grad():
        for batch in batches:
            output = model(batch[:, 1:-1])
            ce = torch.nn.functional.cross_entropy(output.flatten(end_dim=-2), batch[:, 2:].flatten())
            position = batch[:, 0]
            predict = output[:, position - 2, :].argmax(-1)
            loss += float(ce)
            ppl += float(ce.exp())
            acc += float((predict == batch[:, position]).float().mean())
    assert abs(got['loss'] - loss / 3) < 1e-5 and abs(got['ppl'] - ppl / 3) < 1e-3
    assert abs(got['accuracy'] - acc / 3) < 1e-6


def test_tokenizer_is_never_fetched(tmp_path):
    from naive_gpt import loaders
    try:
        loaders.resolve_tokenizer('facebook/opt-1.3b')
        assert False
    except FileNotFoundError as exc:
        assert 'local directory' in str(exc)
    assert isinstance(loaders.resolve_tokenizer('bytes'), loaders.ByteTokenizer)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_mmlu_evaluation_of_an_upgraded_checkpoint_on_the_gpu(tmp_path):
    """script/3-mmlu-evaluate.py end to end on local files: base checkpoint -> four-stage upgrade ->
    tuned adapters on top (strict=False, no adapter key missing) -> MMLU test batches through the HIP
    path (PQ sparse attention + routed FFN) -> finite perplexity, accuracy in [0, 1]; and the tuned
    tables really are the ones evaluated."""
    from naive_gpt import loaders, models, utils
    _write_mmlu(str(tmp_path))
    tok = loaders.ByteTokenizer()
    torch.manual_seed(0)
    config = dict(d_model=128, n_heads=2, n_layers=2, max_length=256, vocab_size=tok.vocab_size,
                  d_feedforward=512, p_dropout=0.0)
    base = models.OPTModel(**config)
    ckpt = str(tmp_path / 'opt-tiny.ckpt')
    utils.save_checkpoint(ckpt, config, base)
    tuned = utils.upgrade_sparse(utils.model_from_checkpoint(ckpt), d_lora=16)
    for name, p in tuned.named_parameters():
        if name.endswith('lora.right.weight'):
            p.data.normal_(0, 0.05)
    spt = str(tmp_path / 'opt-tiny-spt.ckpt')
    utils.save_checkpoint(spt, config, tuned)
    model = utils.load_spt_model(ckpt, spt, d_lora=16, device='cuda')
    for (n1, p1), (n2, p2) in zip(model.named_parameters(), tuned.named_parameters()):
        assert n1 == n2 and torch.equal(p1.cpu(), p2), n1
    dm = loaders.MMLUDataModule(root=str(tmp_path), n_shots=2, seq_length=257, batch_size=4, tokenizer=tok)
    got = utils.evaluate_mmlu(model, dm.test_dataloader(), n_batches=3, device='cuda')
    assert got['batches'] == 3 and got['loss'] == got['loss'] and 0.0 <= got['accuracy'] <= 1.0
    assert 1.0 < got['ppl'] < 1e4
    from naive_gpt import ext
    assert ext.PATH_COUNTS[('attention', 'mfma')] + ext.PATH_COUNTS[('attention', 'fused_gather')] > 0
