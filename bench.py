"""bench.py -- the SPT fine-tune step on MI355X (BASELINE.json configs[2]).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        # no launcher: bench.py starts its N ranks itself
    python bench.py --config llama-7b   # BASELINE.json configs[4] (opt-1.3b: configs[3])

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU; it
starts N child rank processes of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, as
torch.distributed.run would), relays rank 0's single JSON line and exits with the children's code.

Headline workload = BASELINE.json configs[2]: BERT-large dimensions (the reference's
`opt-1024`: d_model 1024, 16 heads x 64, d_ff 4096, script/0-profile.py:16-19), 24 layers,
seq 512, micro-batch 16 per GPU, fp32, four-stage upgraded (lora -> ffn -> mha_v1 -> mha_v2:
LoRA + routed FFN + PQ sparse attention).  One step = `utils.SparseTuner.training_step`
(script/4-sparse-tuning-0.py:66-93 restated): arm the PQ triggers, forward, CE + 1e-2 * PQ
loss, backward, [N > 1: ONE flat RCCL all-reduce of the trainable gradients], clip 1.0,
AdamW -- on resident synthetic tokens.  Data parallel: every rank its own micro-batch
(weak scaling), identical replicas, the all-reduce INSIDE the timed step.

Rank 0 prints ONE JSON line: tokens/s over all ranks and peak HBM, plus
  full / lora  -- the same model and step as a dense full fine-tune and as LoRA only, on the
                  same GPU (the >= 2x / <= 50 % claims; N = 1 only)
  graphed      -- the headline step captured as one HIP graph (SparseTuner.capture), N = 1 only
  block        -- ONE TransformerBlock under the protocol of script/0-profile.py:203-226
                  (fwd + bwd + AdamW on randn[16, 512, 1024]): sparse / full / lora, and the
                  sparse step as a HIP graph
  attention    -- BASELINE.json configs[1] (sparse MHA only, fwd + bwd), last round's headline
  roofline     -- the dominant HIP kernel of the headline step (the split-bf16 grouped GEMM),
                  HIP events around every 7th launch inside the timed steps
  cpu_baseline -- the dense PyTorch-CPU counterpart of the same model step on the host's cores
                  (+ the single-core C oracle of the attention chain), N = 1 only
"""
import argparse
import contextlib
import io
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'spt-proto_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from torch import nn, optim  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s HBM3E (spec)
MFMA_BF16_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16
MFMA_FP32_PEAK_TF = 157.3    # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector rate

# The workloads BASELINE.json names.  bert-large = the reference's `opt-1024`
# (script/0-profile.py:16-19); opt-1.3b / llama-7b = the dimensions script/1-convert.py:202-206
# converts (facebook/opt-1.3b, open_llama_7b), random init -- no checkpoint exists offline.
CONFIGS = {
    'bert-large': dict(family='opt', d_model=1024, n_heads=16, d_ff=4096, layers=24, vocab=30522,
                       seq=512, batch=16,
                       workload='BASELINE.json configs[2]: BERT-large sparse-MHA + routed-FFN full '
                                'fine-tune step'),
    'opt-1.3b': dict(family='opt', d_model=2048, n_heads=32, d_ff=8192, layers=24, vocab=50272,
                     seq=2048, batch=2,
                     workload='BASELINE.json configs[3]: OPT-1.3B sparse-MHA + routed-FFN fine-tune '
                              'step, seq 2048 (long-seq CSR stress)'),
    'llama-7b': dict(family='llama', d_model=4096, n_heads=32, d_ff=11008, layers=32, vocab=32000,
                     seq=2048, batch=1,
                     workload='BASELINE.json configs[4]: LLaMA-7B sparse fine-tune step, seq 2048, '
                              'data parallel with one RCCL all-reduce of the trainable gradients'),
    # plumbing check of the launcher on CPU (tests/test_distributed.py); only with --rehearse-cpu
    'tiny-rehearsal': dict(family='opt', d_model=32, n_heads=2, d_ff=64, layers=1, vocab=64,
                           seq=16, batch=2, workload='launcher rehearsal (no BASELINE config)'),
}
FAMILY = 'opt'
D_MODEL, H, D_FF, LAYERS, VOCAB = 1024, 16, 4096, 24, 30522
S = 512
E = D_MODEL // H
M, C, D = E // 8, 16, 8
Z = S // 8
WORKLOAD = CONFIGS['bert-large']['workload']


def set_config(name):
    """Bind the module-level dimensions every record reads to one of CONFIGS."""
    global FAMILY, D_MODEL, H, D_FF, LAYERS, VOCAB, S, E, M, Z, WORKLOAD
    c = CONFIGS[name]
    FAMILY, D_MODEL, H, D_FF = c['family'], c['d_model'], c['n_heads'], c['d_ff']
    LAYERS, VOCAB, S, WORKLOAD = c['layers'], c['vocab'], c['seq'], c['workload']
    E = D_MODEL // H
    M = E // 8
    Z = S // 8
    return c


# ------------------------------------------------------------------------------- timing
CPU_REHEARSAL = False        # --rehearse-cpu: the launcher / DP plumbing without a GPU (tests)


def _sync():
    if not CPU_REHEARSAL:
        torch.cuda.synchronize()


def timed_loop(fn, steps, warmup, world):
    """W untimed steps, then exactly K steps between barrier + synchronize; max over ranks."""
    for _ in range(warmup):
        fn()
    _sync()
    if world > 1:
        dist.barrier()
    _sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    _sync()
    if world > 1:
        dist.barrier()
    _sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device='cpu' if CPU_REHEARSAL else 'cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


class GemmTimer:
    """HIP events (torch's current stream = the stream the C ABI launches on) around every
    `ext.grouped_gemm` / `ext.grouped_gemm_fused` call: the routed FFN's block GEMMs and every
    frozen LoRA linear are launches of the one `spt::grouped_gemm_kernel`."""

    # Two event records per launch on all 288 launches of a step cost the host 5-8 ms of a
    # 75 ms step (measured: 74.9 ms without, 78-83 ms with), so only every 7th launch is
    # bracketed: 7 is coprime to the 12 launches of a layer, so over the timed steps every
    # call site (the four FFN GEMMs, the eight LoRA-linear GEMMs) is sampled equally often.
    EVERY = 7

    def __init__(self):
        from naive_gpt import ext
        self.ext = ext
        self.enabled = False
        self.seen = 0
        self.pool, self.records = [], []
        self.orig = {name: getattr(ext, name) for name in ('grouped_gemm', 'grouped_gemm_fused')}
        ext.grouped_gemm = self._wrap('grouped_gemm')
        ext.grouped_gemm_fused = self._wrap('grouped_gemm_fused')

    def restore(self):
        for name, fn in self.orig.items():
            setattr(self.ext, name, fn)

    def reserve(self, n_pairs):
        while len(self.pool) < 2 * n_pairs:
            self.pool.append(torch.cuda.Event(enable_timing=True))

    def _wrap(self, name):
        fn, ext = self.orig[name], self.ext

        def timed(a, weight, offsets, n_groups, n, k, *args, **kwargs):
            if not self.enabled:
                return fn(a, weight, offsets, n_groups, n, k, *args, **kwargs)
            self.seen += 1
            if self.seen % self.EVERY:
                return fn(a, weight, offsets, n_groups, n, k, *args, **kwargs)
            if name == 'grouped_gemm':
                gather = kwargs.get('gather', args[3] if len(args) > 3 else None)
                rows = kwargs.get('n_rows') or (gather.numel() if gather is not None else a.size(0))
                rank = 0
            else:
                rows = kwargs.get('n_rows', args[3] if len(args) > 3 else None)
                a2 = kwargs.get('a2')
                rank = a2.size(1) if a2 is not None else 0
            if len(self.pool) < 2:
                self.reserve(256)
            e0, e1 = self.pool.pop(), self.pool.pop()
            e0.record()
            out = fn(a, weight, offsets, n_groups, n, k, *args, **kwargs)
            e1.record()
            self.records.append((e0, e1, 2.0 * rows * n * (k + rank), 3))
            return out
        return timed

    def summary(self):
        if not self.records:
            return None
        ms = np.array([a.elapsed_time(b) for a, b, _, _ in self.records])
        flops = np.array([f for _, _, f, _ in self.records])
        parts = np.array([p for _, _, _, p in self.records])
        total_s = float(ms.sum()) * 1e-3
        scale = self.seen / len(ms)         # sampled launches -> all launches
        return {'calls': self.seen, 'sampled': len(ms), 'total_ms': float(ms.sum()) * scale,
                'avg_us': 1e3 * float(ms.mean()),
                'algorithmic_TFLOPs': float(flops.sum()) / total_s / 1e12,
                'executed_TFLOPs': float((flops * parts).sum()) / total_s / 1e12,
                'flops_per_launch': float(flops.mean())}


# ------------------------------------------------------------------------------- builders
class SDPAAttention(nn.Module):
    """Dense causal attention through the library's fused kernel
    (F.scaled_dot_product_attention, is_causal): the `full_sdpa` / `lora_sdpa` legs -- the dense
    baselines with an attention that, like the sparse path, never materialises the [S, S] scores.
    Same function as layers.VanillaAttention with the causal mask (attention.py:21-31)."""

    def __init__(self, d_head: int):
        super().__init__()
        self.scaling = float(d_head) ** -0.5

    def forward(self, q, k, v, attn_mask=None):
        y = nn.functional.scaled_dot_product_attention(
            q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), is_causal=True, scale=self.scaling)
        return y.transpose(1, 2).contiguous()


def build_model(tuning, dev, layers=LAYERS):
    from naive_gpt import models, utils
    torch.manual_seed(0)
    sdpa = tuning.endswith('_sdpa')
    tuning = tuning.replace('_sdpa', '')
    with torch.device(dev):           # (a 7 B model is built on the GPU, not copied to it)
        cls = models.LLaMAModel if FAMILY == 'llama' else models.OPTModel
        model = cls(d_model=D_MODEL, n_heads=H, n_layers=layers, max_length=S,
                    vocab_size=VOCAB, d_feedforward=D_FF, p_dropout=0.0)
        if tuning == 'lora':
            model = utils.upgrade_sparse(model, d_lora=16, stages=('lora',))
        elif tuning == 'sparse':
            model = utils.upgrade_sparse(model, d_lora=16)
        if sdpa:
            for block in model.decoders:
                block.mha.attn_fn = SDPAAttention(E)
    return model.to(dev)


def build_block(tuning, dev):
    from naive_gpt import layers, utils
    torch.manual_seed(0)
    with torch.device(dev):
        block = layers.TransformerBlock(
            d_model=D_MODEL, n_heads=H, layernorm_fn=nn.LayerNorm(D_MODEL),
            attention_fn=layers.VanillaAttention(d_head=E, p_dropout=0.0),
            feedforward_fn=layers.Feedforward(d_model=D_MODEL, d_feedforward=D_FF,
                                              activation=nn.ReLU(), p_dropout=0.0),
            attention_bias=True, pre_norm=True)
        with contextlib.redirect_stdout(io.StringIO()):
            if tuning == 'lora':
                block = utils.ModuleUpgrader(utils.LoRAHandler(d_lora=16)).visit(block)
            elif tuning == 'sparse':
                for stage in ('lora', 'ffn', 'mha_v1', 'mha_v2'):
                    block = utils.ModuleUpgrader(
                        utils.SparseLoRAHandler(d_lora=16, stage=stage)).visit(block)
    return block.to(dev)


def model_record(tuning, args, world, rank, dev, gemm_timer=None):
    """`SparseTuner.training_step` of the 24-layer model; at world > 1 the tuner broadcasts
    rank 0's replica and all-reduces the trainable gradients inside every step."""
    from naive_gpt import utils
    if not CPU_REHEARSAL:
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
    model = build_model(tuning, dev, args.layers)
    tuner = utils.SparseTuner(model)
    if not CPU_REHEARSAL:
        # the peak is that of the STEPS on top of the resident model, not of the model's
        # construction (the upgrader's transient copies: 14 GB at the LLaMA-7B dimensions)
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    N = args.batch

    def step():
        # batch[:, 0] is the MMLU answer position, [:, 1:-1] the input, [:, 2:] the target
        batch = torch.randint(3, VOCAB, [N, S + 2], device=dev, generator=gen)
        tuner.training_step(batch, pq_loss=(tuning == 'sparse'))

    rec = {}
    if world > 1:
        # step 0 of the run, outside the timed region: every rank must clip by the SAME norm (the
        # norm is taken after the exchange) -- what proves that the all-reduce averaged the ranks'
        # different micro-batches and that the replicas started identical
        step()
        norms = [torch.zeros([], device=dev) for _ in range(world)]
        dist.all_gather(norms, tuner.last_grad_norm.detach().float().reshape([]))
        norms = [float(t) for t in norms]
        assert all(abs(n - norms[0]) <= 1e-6 * abs(norms[0]) for n in norms), \
            'clip norms differ over the ranks: {}'.format(norms)
        rec['clip_norm_step0'] = norms[0]
        rec['clip_norm_equal_on_all_ranks'] = True
    timed_loop(step, 0, args.warmup, world)
    if gemm_timer is not None:
        gemm_timer.reserve(2 * args.layers * args.steps + 8)
        gemm_timer.enabled = True
    if world > 1:
        tuner.allreduce_every = 2                  # HIP events around every 2nd exchange, timed region
        tuner.allreduce_events.clear()
    dt = timed_loop(step, args.steps, 0, world)
    if gemm_timer is not None:
        gemm_timer.enabled = False
    tokens = N * S * world * args.steps
    trainable = sum(p.numel() for p in tuner.params)
    rec.update({'value': tokens / dt, 'unit': 'tokens/s', 'ms_per_step': 1e3 * dt / args.steps,
                'peak_hbm_gb': None if CPU_REHEARSAL else torch.cuda.max_memory_allocated() / 1e9,
                'trainable_params': trainable,
                'total_params': sum(p.numel() for p in model.parameters())})
    if world > 1:
        nbytes = 4 * (tuner._flat.numel() if tuner._flat is not None else trainable)
        rec['allreduce_bytes_per_step'] = nbytes
        if tuner.allreduce_events:
            ms = float(np.mean([a.elapsed_time(b) for a, b in tuner.allreduce_events]))
            t = torch.tensor([ms], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t.item())
            rec['allreduce_ms'] = ms
            # ring / tree-independent "bus bandwidth" of an all-reduce: 2 (N - 1) / N of the bytes
            rec['allreduce_bus_gbs'] = 2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9
            rec['allreduce_sampled'] = len(tuner.allreduce_events)
    del model, tuner
    if not CPU_REHEARSAL:
        _release()
    return rec


def graphed_model_record(args, dev):
    """The same sparse step captured as ONE HIP graph (`SparseTuner.capture`): what the host's
    ~3,900 launches per step cost.  N = 1 only."""
    from naive_gpt import utils
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    model = build_model('sparse', dev, args.layers)
    tuner = utils.SparseTuner(model)
    gen = torch.Generator(device=dev).manual_seed(1)
    N = args.batch
    tuner.capture([N, S + 2], pq_loss=True, warmup=max(args.warmup, 3),
                  example=torch.randint(3, VOCAB, [N, S + 2], device=dev, generator=gen))

    def step():
        tuner.training_step(torch.randint(3, VOCAB, [N, S + 2], device=dev, generator=gen))

    dt = timed_loop(step, args.steps, 2, 1)
    rec = {'value': N * S * args.steps / dt, 'unit': 'tokens/s', 'ms_per_step': 1e3 * dt / args.steps,
           'peak_hbm_gb': torch.cuda.max_memory_allocated() / 1e9,
           'what': 'SparseTuner.capture(): forward, backward, clip and AdamW (capturable) of the '
                   'same step replayed as one HIP graph'}
    del model, tuner
    _release()
    return rec


def graphed_block_record(args, dev):
    """The sparse block step of `block_record` as one HIP graph (x = randn inside the graph)."""
    torch.cuda.empty_cache()
    block = build_block('sparse', dev)
    params = [p for p in block.parameters() if p.requires_grad]
    opt = optim.AdamW(params, lr=torch.tensor(1e-4, device=dev), weight_decay=1e-2, capturable=True,
                      fused=True)
    N = args.batch

    def step():
        x = torch.randn([N, S, D_MODEL], device=dev, requires_grad=True)
        block(x, attn_mask=None).sum().backward()
        opt.step()
        block.zero_grad()

    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(5):
            step()
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    steps = max(args.steps, 20)
    dt = timed_loop(graph.replay, steps, 20, 1)
    rec = {'value': N * S * steps / dt, 'unit': 'tokens/s', 'ms_per_step': 1e3 * dt / steps}
    del block, opt, graph
    _release()
    return rec


def _release():
    """Nothing of one record may sit in the next one's peak: the operand-image caches of
    naive_gpt.ext hold the last activations they saw."""
    import gc
    from naive_gpt import ext
    from naive_gpt.layers.tuning import recompute
    ext.drop_images()
    recompute.release()
    gc.collect()
    torch.cuda.empty_cache()


def block_record(tuning, args, dev):
    """script/0-profile.py:203-226: x = randn[N, S, d]; y = block(x); y.sum().backward();
    AdamW step; zero_grad -- one TransformerBlock, triggers never armed."""
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    block = build_block(tuning, dev)
    params = [p for p in block.parameters() if p.requires_grad]
    opt = optim.AdamW(params, lr=1e-4, weight_decay=1e-2, fused=True)   # (all three variants alike)
    mask = None if tuning == 'sparse' else torch.full([S, S], float('-inf'), device=dev).triu(1)
    N = args.batch

    def step():
        x = torch.randn([N, S, D_MODEL], device=dev, requires_grad=True)
        block(x, attn_mask=mask).sum().backward()
        opt.step()
        block.zero_grad()

    steps = max(args.steps, 20)
    dt = timed_loop(step, steps, 20, 1)       # the reference protocol's 20 warm-up steps
    rec = {'value': N * S * steps / dt, 'unit': 'tokens/s', 'ms_per_step': 1e3 * dt / steps,
           'peak_hbm_gb': torch.cuda.max_memory_allocated() / 1e9,
           'trainable_params': sum(p.numel() for p in params)}
    del block, opt
    _release()
    return rec


# ------------------------------------------------------------- configs[1]: attention only
def attention_record(args, dev, dtype=torch.float32):
    """BASELINE.json configs[1]: `SparseVanillaAttentionV2` fwd + bwd on randn[N, S, H, E]
    (cdist / lookup / sddmm / softmax / spmm), its dominant kernel against the HBM roofline,
    dense causal attention beside it.  dtype bfloat16: q, k, v (and y, the gradients) stored
    bf16, both for the sparse layer and for its dense counterpart."""
    from naive_gpt import ext, layers
    N = args.batch
    torch.manual_seed(0)
    attn = layers.SparseVanillaAttentionV2(d_head=E, d_codeword=D, n_codewords=C,
                                           p_dropout=0.0).to(dev)
    q, k, v = [torch.randn([N, S, H, E], device=dev).to(dtype).requires_grad_(True) for _ in range(3)]
    eb = 2 if dtype == torch.bfloat16 else 4
    name = 'bf16 storage' if dtype == torch.bfloat16 else 'fp32'

    events = []
    orig_bwd = ext.attention_mfma_backward

    def timed_bwd(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_bwd(*a, **kw)
        e1.record()
        events.append((e0, e1))
        return out

    def step():
        for t in (q, k, v):
            t.grad = None
        attn(q, k, v, attn_mask=None).sum().backward()

    torch.cuda.reset_peak_memory_stats()
    timed_loop(step, 0, 300, 1)               # < 1 ms steps: leave the idle clocks first
    steps = 200
    dt = timed_loop(step, steps, 0, 1)
    peak = torch.cuda.max_memory_allocated() / 1e9
    ext.attention_mfma_backward = timed_bwd
    timed_loop(step, 50, 0, 1)
    ext.attention_mfma_backward = orig_bwd
    B = N * H
    rec = {'workload': 'BASELINE.json configs[1]: BERT-large sparse-MHA only, fwd+bwd, ' + name,
           'dtype': 'bf16' if dtype == torch.bfloat16 else 'f32',
           'value': N * S * steps / dt, 'unit': 'tokens/s', 'ms_per_step': 1e3 * dt / steps,
           'peak_hbm_gb': peak}
    if events:
        us = 1e3 * float(np.mean([a.elapsed_time(b) for a, b in events]))
        nbytes = (8 * S * E * eb + S * Z * 4) * B      # q k v dY y read, 3 gradients written, CSR
        rec['roofline'] = {'kernel': 'attention_mfma_backward (two launches)', 'bound': 'hbm',
                           'achieved': nbytes / (us * 1e-6) / 1e9, 'peak': HBM_PEAK_GBS,
                           'unit': 'GB/s', 'frac': nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                           'avg_us': us, 'bytes_per_launch': nbytes}
    del attn
    dense = layers.VanillaAttention(d_head=E, p_dropout=0.0).to(dev)
    mask = torch.full([S, S], float('-inf'), device=dev, dtype=dtype).triu(1)

    def dense_step():
        for t in (q, k, v):
            t.grad = None
        dense(q, k, v, attn_mask=mask).sum().backward()

    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    ddt = timed_loop(dense_step, 50, 10, 1)
    rec['dense_ms_per_step'] = 1e3 * ddt / 50
    rec['speedup_vs_dense'] = rec['dense_ms_per_step'] / rec['ms_per_step']
    rec['peak_hbm_vs_dense'] = peak / (torch.cuda.max_memory_allocated() / 1e9)
    return rec


# ------------------------------------------------------------------------------ CPU legs
def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or 'unknown'


def cpu_oracle(n_seq):
    """The oracle's single-core attention chain on `n_seq` sequences x 16 heads (fwd + bwd)."""
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    B = n_seq * H
    q, k, v, dy = [rng.standard_normal([B, S, E]).astype(np.float32) for _ in range(4)]
    table = rng.standard_normal([M, C, D]).astype(np.float32)
    indptr = (np.arange(S + 1) * Z).astype(np.int32)

    def codes(z):
        zf = np.ascontiguousarray(z.reshape(B * S, M, D).transpose(1, 0, 2))
        return np.ascontiguousarray(O.cdist_forward(zf, table, False)[1].T).reshape(B, S, M)

    t0 = time.perf_counter()
    idx = O.lookup_forward(codes(q), codes(k), 8).reshape(B, -1)
    raw = O.sddmm_forward(indptr, idx, q, k) * np.float32(E ** -0.5)
    a = O.softmax_forward(indptr, idx, np.clip(raw, -10, 10))
    O.spmm_forward(False, indptr, idx, a, v)
    da = O.sddmm_forward(indptr, idx, dy, v)
    O.spmm_forward(True, indptr, idx, a, dy)
    dr = O.softmax_backward(indptr, idx, a, da)
    dr = np.where(np.abs(raw) < 10, dr * np.float32(E ** -0.5), 0).astype(np.float32)
    O.spmm_forward(False, indptr, idx, dr, k)
    O.spmm_forward(True, indptr, idx, dr, q)
    dt = time.perf_counter() - t0
    return {'value': n_seq * S / dt, 'unit': 'tokens/s (attention chain only, fwd+bwd)',
            'cores': 1, 'sample': '{} sequences x {} heads x seq {} through oracle/spt_oracle.c, '
                                  '{:.1f} s'.format(n_seq, H, S, dt)}


def cpu_baseline(args):
    """SURVEY 8(d) / BASELINE.md 3: the build's own dense PyTorch-CPU counterpart of the
    headline step -- the same 24-layer model as a dense full fine-tune (the only form the
    reference can run without its CUDA extension), `SparseTuner.training_step` on all host
    threads -- on a bounded sample: ONE step of `cpu_batch` sequences."""
    from naive_gpt import utils
    threads = torch.get_num_threads()
    torch.manual_seed(0)
    model = build_model('full', 'cpu', args.layers)
    tuner = utils.SparseTuner(model)
    gen = torch.Generator().manual_seed(1)
    tuner.training_step(torch.randint(3, VOCAB, [1, 34], generator=gen), pq_loss=False)  # pools
    batch = torch.randint(3, VOCAB, [args.cpu_batch, S + 2], generator=gen)
    t0 = time.perf_counter()
    tuner.training_step(batch, pq_loss=False)
    dt = time.perf_counter() - t0
    del model, tuner
    rec = {'value': args.cpu_batch * S / dt, 'unit': 'tokens/s', 'cores': threads, 'kind': 'port',
           'sample': 'one SparseTuner.training_step (fwd + bwd + clip + AdamW) of the dense '
                     '{}-layer model, {} sequences x seq {}, PyTorch CPU fp32, {:.1f} s'.format(
                         args.layers, args.cpu_batch, S, dt),
           'cpu_model': cpu_model_name(), 'os_cpu_count': os.cpu_count(),
           'torch_num_threads': threads}
    try:
        rec['oracle_1core'] = cpu_oracle(args.cpu_seqs)
    except Exception as exc:                    # the baseline line must not die with the extra
        rec['oracle_1core'] = {'error': repr(exc)}
    try:
        rec['c1_dense_block_forward'] = cpu_c1()
    except Exception as exc:
        rec['c1_dense_block_forward'] = {'error': repr(exc)}
    return rec


def cpu_c1():
    """BASELINE.json configs[0] / BASELINE.md 3 "C1": the dense TransformerBlock at BERT-base
    dimensions (d 768, 12 heads, d_ff 3072), `randn[8, 128, 768]`, forward only, PyTorch CPU -- the
    reference's own CPU-runnable case (plumbing; BASELINE.md 4 quotes 36.6 ms for the imported
    reference on the 8-core survey container).  3 warm-up + 10 timed calls, median."""
    from naive_gpt import layers
    torch.manual_seed(0)
    d, heads, d_ff, n, s = 768, 12, 3072, 8, 128
    block = layers.TransformerBlock(
        d_model=d, n_heads=heads, layernorm_fn=nn.LayerNorm(d),
        attention_fn=layers.VanillaAttention(d_head=d // heads, p_dropout=0.0),
        feedforward_fn=layers.Feedforward(d_model=d, d_feedforward=d_ff, activation=nn.ReLU(),
                                          p_dropout=0.0),
        attention_bias=True, pre_norm=True)
    x = torch.randn([n, s, d])
    mask = torch.full([s, s], float('-inf')).triu(1)
    times = []
    with torch.no_grad():
        for i in range(13):
            t0 = time.perf_counter()
            block(x, attn_mask=mask)
            if i >= 3:
                times.append(time.perf_counter() - t0)
    ms = 1e3 * float(np.median(times))
    return {'value': n * s / (ms * 1e-3), 'unit': 'tokens/s', 'ms_per_call': ms,
            'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': 'dense TransformerBlock d 768 / 12 heads / d_ff 3072, randn[8, 128, 768], '
                      'forward, PyTorch CPU fp32, median of 10 calls'}


def traffic_file():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')))
    return os.path.relpath(files[-1], ROOT) if files else None


def measured_traffic(kernel_prefix):
    """HBM bytes per launch from the committed PMC passes (profiles/*_traffic.json:
    FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE), launch-weighted over the
    instantiations of the kernel; None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')))
    if not files:
        return None
    table = json.load(open(files[-1]))
    rows = [v for k, v in table.items() if k.startswith(kernel_prefix)]
    if not rows:
        return None
    calls = sum(r.get('launches', 1) for r in rows)
    return sum(r['hbm_bytes_per_launch'] * r.get('launches', 1) for r in rows) / calls


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def launch_ranks(n):
    """`--gpus N` without a launcher: start the N rank processes of this same command line as
    CHILDREN (this parent has made no GPU call and makes none: a process that has initialised
    the GPU must never be replaced or forked into ranks), with the environment
    torch.distributed.run would give them; relay rank 0's stdout (the one JSON line), send the
    other ranks' stdout to stderr, kill the rest when a rank fails, return the worst exit code."""
    import subprocess
    port = os.environ.get('MASTER_PORT') or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      text=True))

    def relay():            # rank 0: JSON lines to stdout, library chatter (gloo / RCCL) to stderr
        for line in procs[0].stdout:
            out = sys.stdout if line.lstrip().startswith('{') else sys.stderr
            out.write(line)
            out.flush()

    import threading
    pump = threading.Thread(target=relay, daemon=True)
    pump.start()
    rc, alive = 0, list(procs)
    while alive:
        time.sleep(0.2)
        for proc in list(alive):
            code = proc.poll()
            if code is None:
                continue
            alive.remove(proc)
            if code != 0:
                rc = rc or code
                for other in alive:            # a dead rank leaves the others in a collective
                    other.terminate()
    pump.join(timeout=10)
    return rc


def main():
    global CPU_REHEARSAL
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='default 20 (bert-large) / 8')
    ap.add_argument('--warmup', type=int, default=None, help='default 10 (bert-large) / 3')
    ap.add_argument('--config', default='bert-large', choices=sorted(CONFIGS),
                    help='bert-large = BASELINE configs[2] (the headline), opt-1.3b = configs[3], '
                         'llama-7b = configs[4]')
    ap.add_argument('--batch', type=int, default=None, help='micro-batch per GPU (default: the config\'s)')
    ap.add_argument('--layers', type=int, default=None)
    ap.add_argument('--no-baselines', action='store_true', help='skip full / lora')
    ap.add_argument('--no-block', action='store_true')
    ap.add_argument('--no-attention', action='store_true')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='skip the HIP-graph records')
    ap.add_argument('--no-gemm-events', action='store_true')
    ap.add_argument('--only', action='store_true', help='the headline step alone')
    ap.add_argument('--backend', default='nccl', help="process-group backend (nccl = RCCL)")
    ap.add_argument('--one-device', action='store_true',
                    help='rehearsal of the N > 1 path on a one-GPU box: every rank on cuda:0 '
                         '(use with --backend gloo); not a measurement')
    ap.add_argument('--rehearse-cpu', action='store_true',
                    help='launcher / DP plumbing check without a GPU (tests/test_distributed.py): the '
                         'DENSE model of the config on CPU tensors over gloo.  The sparse path has no '
                         'CPU form; the line says "rehearsal" and is not a measurement')
    ap.add_argument('--cpu-batch', type=int, default=2)
    ap.add_argument('--cpu-seqs', type=int, default=16)
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus))         # before anything touches a GPU

    if args.config == 'tiny-rehearsal' and not args.rehearse_cpu:
        ap.error('--config tiny-rehearsal is the launcher test\'s model: use it with --rehearse-cpu')
    cfg = set_config(args.config)
    headline = args.config == 'bert-large'
    if args.steps is None:
        args.steps = 20 if headline else 8
    if args.warmup is None:
        args.warmup = 10 if headline else 3
    if args.batch is None:
        args.batch = cfg['batch']
    if args.layers is None:
        args.layers = cfg['layers']
    if args.only or args.rehearse_cpu:
        args.no_baselines = args.no_block = args.no_attention = args.no_cpu = args.no_graph = True
    if not headline:
        # the block / attention / graph / CPU records describe the headline workload
        # (tools/bench_block.py, tools/bench_long.py measure them at the other dimensions)
        args.no_block = args.no_attention = args.no_cpu = args.no_graph = True

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'WORLD_SIZE={} but --gpus {}'.format(world, args.gpus)
    if args.rehearse_cpu:
        CPU_REHEARSAL = True
        args.backend = 'gloo'
        args.no_gemm_events = True
        dev = torch.device('cpu')
    else:
        if args.one_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)

    if args.rehearse_cpu:
        rec = model_record('full', args, world, rank, dev)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({
                'rehearsal': 'CPU tensors over gloo, DENSE model: launcher and data-parallel '
                             'plumbing only, NOT a measurement of the sparse path',
                'metric': 'fine-tune tokens/sec', 'value': rec['value'], 'unit': 'tokens/s',
                'n_gpus': 0, 'ranks': world, 'backend': 'gloo', 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': rec['ms_per_step'],
                'config': {'workload': WORKLOAD, 'n_layers': args.layers, 'seq_len': S,
                           'micro_batch_per_gpu': args.batch},
                'clip_norm_equal_on_all_ranks': rec.get('clip_norm_equal_on_all_ranks')}))
        return

    from naive_gpt import ext
    ext.load_library()        # fail loudly when the HIP library is missing

    gemm_timer = None if args.no_gemm_events else GemmTimer()
    sparse = model_record('sparse', args, world, rank, dev, gemm_timer)
    gemm = gemm_timer.summary() if gemm_timer is not None else None
    if gemm_timer is not None:
        gemm_timer.records.clear()
        gemm_timer.pool.clear()
        gemm_timer.restore()

    N = args.batch
    result = {
        'metric': 'fine-tune tokens/sec + peak HBM GB, BERT-large seq=512' if headline else
                  'fine-tune tokens/sec + peak HBM GB, {} dims seq={}'.format(args.config, S),
        'value': sparse['value'], 'unit': 'tokens/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': sparse['ms_per_step'],
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {
            'workload': WORKLOAD + ' (SparseTuner.training_step: PQ triggers armed, fwd, CE + '
                        '1e-2 PQ loss, bwd, gradient all-reduce, clip 1.0, AdamW), {} layers, '
                        'four-stage upgraded, random init'.format(args.layers),
            'name': args.config, 'family': FAMILY,
            'd_model': D_MODEL, 'n_heads': H, 'd_head': E, 'd_ff': D_FF, 'n_layers': args.layers,
            'vocab': VOCAB, 'seq_len': S, 'micro_batch_per_gpu': N, 'global_batch': N * world,
            'nnz_per_row': Z, 'pq': [M, C, D], 'ffn_blocks': 4, 'ffn_top_k': 2, 'd_lora': 16,
            'parallelism': 'dp{}'.format(world),
            'arithmetic': 'fp32 tensors; dense and attention products = 3 bf16 MFMAs on hi/lo-'
                          'split fp32 operands, fp32 accumulation (ReLU pre-activations within '
                          'the split error of zero recomputed in fp32); PQ codes / top-k '
                          'indices exact.  The dense full fine-tune leg (`full`) runs its linears '
                          'on the library fp32 GEMM; the LoRA leg runs its frozen linears on the '
                          'same split-bf16 engine as the sparse step'},
        'peak_hbm_gb': sparse['peak_hbm_gb'],
        'trainable_params': sparse['trainable_params'], 'total_params': sparse['total_params'],
    }
    if world > 1:
        result['ranks'] = dist.get_world_size()
        result['backend'] = dist.get_backend()
        for key in ('allreduce_bytes_per_step', 'allreduce_ms', 'allreduce_bus_gbs', 'allreduce_sampled',
                    'clip_norm_step0', 'clip_norm_equal_on_all_ranks'):
            if key in sparse:
                result[key] = sparse[key]

    if rank == 0 and gemm is not None:
        per_step = gemm['calls'] / args.steps
        result['roofline'] = {
            'kernel': 'spt::grouped_gemm_kernel (routed-FFN block GEMMs + every frozen LoRA '
                      'linear of the step)',
            'bound': 'mfma', 'achieved': gemm['algorithmic_TFLOPs'], 'peak': MFMA_BF16_PEAK_TF,
            'unit': 'TFLOP/s', 'frac': gemm['algorithmic_TFLOPs'] / MFMA_BF16_PEAK_TF,
            'frac_algorithmic': gemm['algorithmic_TFLOPs'] / MFMA_BF16_PEAK_TF,
            'executed_TFLOPs': gemm['executed_TFLOPs'],
            'frac_executed': gemm['executed_TFLOPs'] / MFMA_BF16_PEAK_TF,
            'frac_of_fp32_mfma_peak': gemm['algorithmic_TFLOPs'] / MFMA_FP32_PEAK_TF,
            'traffic': measured_traffic('spt::grouped_gemm'),
            'traffic_source': 'bytes per launch from the committed PMC passes ({}), NOT measured in '
                              'this run'.format(traffic_file() or 'none committed'),
            'what': 'achieved / frac = ALGORITHMIC fp32 flops, 2 * rows * n * (k + r) per launch, / '
                    'HIP-event time of every 7th launch inside the timed steps, against the dense '
                    'bf16 MFMA peak; executed = the bf16 MFMA flops the kernel issues for them (3 '
                    'per fp32 product, hi/lo split) = matrix-pipe utilisation (the split passes '
                    'that feed the image path are separate launches, not counted here)',
            'algorithmic_TFLOPs': gemm['algorithmic_TFLOPs'],
            'flops_per_launch': gemm['flops_per_launch'], 'avg_us': gemm['avg_us'],
            'calls_per_step': per_step, 'ms_per_step': gemm['total_ms'] / args.steps,
            'share_of_step': gemm['total_ms'] / args.steps / sparse['ms_per_step']}

    single = world == 1
    if single and not args.no_baselines:
        for tuning in ('full', 'lora'):
            result[tuning] = model_record(tuning, args, world, rank, dev)
        result['speedup_vs_dense'] = result['value'] / result['full']['value']
        result['peak_hbm_vs_dense'] = result['peak_hbm_gb'] / result['full']['peak_hbm_gb']
        result['speedup_vs_lora'] = result['value'] / result['lora']['value']
        result['peak_hbm_vs_lora'] = result['peak_hbm_gb'] / result['lora']['peak_hbm_gb']
        # the same two baselines with a fused dense attention (F.scaled_dot_product_attention):
        # no [S, S] score tensors, so the memory ratio is quoted against that too
        for tuning in ('full_sdpa', 'lora_sdpa'):
            try:
                result[tuning] = model_record(tuning, args, world, rank, dev)
                base = tuning.split('_')[0]
                result['speedup_vs_{}_sdpa'.format(base)] = result['value'] / result[tuning]['value']
                result['peak_hbm_vs_{}_sdpa'.format(base)] = result['peak_hbm_gb'] / result[tuning]['peak_hbm_gb']
            except Exception as exc:
                result[tuning] = {'error': repr(exc)}
        result['baselines'] = (
            'full = dense full fine-tune: every nn.Linear on the LIBRARY fp32 GEMM, naive '
            'softmax(QK^T + mask)V attention, weight gradients + AdamW on all 365 M parameters; '
            'lora = frozen base + rank-16 adapters: frozen linears on the SAME split-bf16 GEMM engine '
            'as the sparse step, naive dense attention -- speedup_vs_lora is the same-engine, '
            'same-trainable-set comparison (what sparsity itself buys); *_sdpa = the same two with '
            "torch's fused scaled_dot_product_attention.  speedup_vs_dense / speedup_vs_full_sdpa "
            'compare across DIFFERENT GEMM engines (and trainable sets); the <= 50 % memory target is met '
            'only against the naive-attention baselines (peak_hbm_vs_lora_sdpa is the same-engine figure)')
        result['speedup_vs_dense_note'] = 'different GEMM engine: library fp32 GEMM vs split-bf16 (see baselines)'
        result['same_engine'] = {'speedup_vs_lora': result['speedup_vs_lora'],
                                 'peak_hbm_vs_lora': result['peak_hbm_vs_lora'],
                                 'speedup_vs_lora_sdpa': result.get('speedup_vs_lora_sdpa'),
                                 'peak_hbm_vs_lora_sdpa': result.get('peak_hbm_vs_lora_sdpa')}
    if single and not args.no_block:
        blk = {t: block_record(t, args, dev) for t in ('sparse', 'full', 'lora')}
        blk['what'] = ('one TransformerBlock, protocol of script/0-profile.py:203-226: fwd + bwd '
                       '+ AdamW on randn[{}, {}, {}], triggers never armed'.format(N, S, D_MODEL))
        blk['speedup_vs_dense'] = blk['sparse']['value'] / blk['full']['value']
        blk['peak_hbm_vs_dense'] = blk['sparse']['peak_hbm_gb'] / blk['full']['peak_hbm_gb']
        result['block'] = blk
    if single and not args.no_attention:
        result['attention'] = attention_record(args, dev)
        try:          # the same workload with bf16 storage (configs[1] "fp32 and bf16")
            result['attention_bf16'] = attention_record(args, dev, torch.bfloat16)
        except Exception as exc:
            result['attention_bf16'] = {'error': repr(exc)}
    if single and not args.no_graph:
        # the HIP-graph records come last (a captured graph's memory pool outlives its record)
        # and must never cost the line
        try:
            result['graphed'] = graphed_model_record(args, dev)
            result['graphed']['speedup_vs_eager'] = result['graphed']['value'] / result['value']
        except Exception as exc:
            result['graphed'] = {'error': repr(exc)}
        if 'block' in result:
            try:
                result['block']['sparse_graphed'] = graphed_block_record(args, dev)
            except Exception as exc:
                result['block']['sparse_graphed'] = {'error': repr(exc)}
    if rank == 0 and single and not args.no_cpu:
        result['cpu_baseline'] = cpu_baseline(args)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == '__main__':
    main()
