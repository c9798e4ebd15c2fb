"""Replay-after-eager divergence: reproduce with the twins of
tests/test_models.py::test_captured_step_replays_the_eager_step (n_seq 16), interleaving the eager
twin's steps with the captured twin's replays WITHOUT host synchronisation, under one variant:

    MODE=plain      as the tuner does it (copy_ of the batch, replay, clone of the loss)
    MODE=sync       torch.cuda.synchronize() in front of every replay
    MODE=event      an event recorded behind the eager work and waited for in front of the replay
    MODE=kcopy      the batch goes into the graph's input through an elementwise KERNEL, not copy_
    MODE=nocopy     the batches are written into the graph input ahead of time (no eager op at all
                    between the twin's eager step and the replay)
    DOT=path        also dump the captured graph (hipGraphDebugDotPrint) and summarise its nodes
"""
import os, re, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import models, utils

MODE = os.environ.get('MODE', 'plain')
DOT = os.environ.get('DOT')
n_seq = int(os.environ.get('NSEQ', 16))
config = dict(d_model=1024, n_heads=16, n_layers=2, max_length=256, vocab_size=512,
              d_feedforward=4096, p_dropout=0.0)


def build():
    torch.manual_seed(3)
    model = models.OPTModel(**config)
    model = utils.upgrade_sparse(model, d_lora=16)
    for name, p in model.named_parameters():
        if name.endswith('lora.right.weight'):
            p.data.normal_(0, 0.02)
    return utils.SparseTuner(model.cuda())


if DOT:
    _Graph = torch.cuda.CUDAGraph

    class DebugGraph(_Graph):
        def __new__(cls, *a, **k):
            g = super().__new__(cls)
            g.enable_debug_mode()
            return g
    torch.cuda.CUDAGraph = DebugGraph

gen = torch.Generator().manual_seed(9)
batches = [torch.randint(3, 512, [n_seq, 258], generator=gen).cuda() for _ in range(3)]
warm = torch.randint(3, 512, [n_seq, 258], generator=gen).cuda()
eager, graphed = build(), build()
eager.use_capturable_optimizer()
graphed.capture(batches[0].shape, pq_loss=True, warmup=3, example=warm)
if DOT:
    graphed._graph.debug_dump(DOT)
    text = open(DOT).read()
    nodes = re.findall(r'^\s*"?([\w.]+)"?\s*\[(.*?)\];', text, re.M | re.S)
    edges = re.findall(r'^\s*"?([\w.]+)"?\s*->\s*"?([\w.]+)"?', text, re.M)
    kinds = collections.Counter()
    label = {}
    for name, attrs in nodes:
        m = re.search(r'label="([^"]*)"', attrs, re.S)
        lab = m.group(1) if m else ''
        label[name] = lab
        kind = 'memset' if re.search('memset', lab, re.I) else 'memcpy' if re.search('memcpy', lab, re.I) \
            else 'kernel' if lab else 'other'
        kinds[kind] += 1
    indeg = collections.Counter(b for _, b in edges)
    outdeg = collections.Counter(a for a, _ in edges)
    roots = [n for n in label if indeg[n] == 0]
    leaves = [n for n in label if outdeg[n] == 0]
    print('graph: nodes', len(label), dict(kinds), 'edges', len(edges), 'roots', len(roots), 'leaves', len(leaves))
    for r in roots[:10]:
        print('  root', r, label[r][:100].replace('\n', ' '))
    multi_in = sum(1 for n in label if indeg[n] > 1)
    multi_out = sum(1 for n in label if outdeg[n] > 1)
    print('  nodes with more than one predecessor', multi_in, 'successor', multi_out)
for _ in range(3):
    eager.training_step(warm, pq_loss=True)
torch.cuda.synchronize()


def apart():
    return sorted(((float((pe - pg).abs().max()), n)
                   for (n, pe), pg in zip(eager.model.named_parameters(), graphed.model.parameters())
                   if pe.requires_grad and not torch.allclose(pe, pg, rtol=1e-6, atol=1e-7)), reverse=True)


def replay(b):
    g = graphed
    if MODE == 'sync':
        torch.cuda.synchronize()
    if MODE == 'event':
        ev = torch.cuda.Event()
        ev.record()
        torch.cuda.current_stream().wait_event(ev)
    if MODE == 'kcopy':
        torch.add(b, 0, out=g._graph_batch)
    elif MODE != 'nocopy':
        g._graph_batch.copy_(b)
    g._graph.replay()
    return g._graph_loss if MODE == 'nocopy' else g._graph_loss.clone()


print('after capture + warm-up: apart', len(apart()))
seq = batches + batches
if MODE == 'nocopy':
    # one batch only, already in place: eager twin takes the same batch six times
    graphed._graph_batch.copy_(batches[0])
    torch.cuda.synchronize()
    seq = [batches[0]] * 6
for i, b in enumerate(seq):
    eager.training_step(b, pq_loss=True)
    replay(b)
torch.cuda.synchronize()
d = apart()
print('MODE', MODE, 'interleaved, no host sync: params apart', len(d), d[:3])
