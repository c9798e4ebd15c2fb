"""bench.py -- throughput of the SPT sparse-attention hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: BERT-large dimensions, sparse MHA only
(cdist -> lookup -> sddmm -> softmax -> spmm and their backward), seq 512, 16 heads
x 64, micro-batch 16 per GPU, fp32 (the reference's only dtype).  One step =
`y = attn(q, k, v); y.sum().backward()` on resident synthetic tensors (protocol of
script/0-profile.py:203-226).  Data parallel: every rank runs its own micro-batch
(weak scaling); trainable gradients (the PQ codebook, present when --trigger arms the
PQ loss as script/4-sparse-tuning-0.py:71-78 does) are all-reduced over RCCL.

Rank 0 prints ONE JSON line: tokens/s over all ranks, plus
  roofline     -- the dominant HIP kernel: algorithmic bytes / its HIP-event time
  cpu_baseline -- the CPU oracle (single core port) on a bounded sample, N=1 only
  dense        -- dense causal attention on the same GPU (the >=2x / <=50% claim)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'spt-proto_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s HBM3E (spec)
MFMA_BF16_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16
# v_mfma_f32_32x32x16_bf16 per 32 x 32 tile (mfma_attention.hip): forward D + PV; backward
# rows kernel D + dP + dQ, keys kernel D + dP + dV + dK
MFMA_PER_TILE = {'attention_mfma_forward': 24, 'attention_mfma_backward': 36 + 48}


S, H, E = 512, 16, 64   # BERT-large: d_model 1024 = 16 heads x 64
M, C, D = E // 8, 16, 8
Z = S // 8


def algorithmic_bytes(op: str, B: int) -> int:
    """SURVEY.md 8(d) per-(batch*head) figures x B slices handled by one launch."""
    per = {
        'cdist_encode': S * E * 4 + S * M * 4,
        'pq_encode_heads': S * E * 4 + S * M * 4,
        'softmax_backward_clamped': 5 * S * Z * 4,
        'cdist_forward_cuda': S * E * 4 + S * M * 4 + S * M * C * 4,
        'cdist_backward_cuda': S * E * 4 + S * M * C * 4 + S * E * 4,
        'lookup_forward_cuda': 2 * S * M * 4 + S * Z * 4,
        'sddmm_forward_cuda': 2 * S * E * 4 + 2 * S * Z * 4,
        'spmm_forward_cuda': 2 * S * Z * 4 + 2 * S * E * 4,
        'spmm_transposed': 2 * S * Z * 4 + 2 * S * E * 4,
        'csr_transpose': 3 * S * Z * 4,
        'softmax_forward_cuda': 3 * S * Z * 4,
        'softmax_backward_cuda': 4 * S * Z * 4,
        'pq_loss_forward': S * E * 4,            # z read once
        'pq_loss_backward': 2 * S * E * 4,       # z read, grad_z written
        # q, k, v read, y written, indices read, scores + probabilities written
        'sparse_attention_forward': 4 * S * E * 4 + 3 * S * Z * 4,
        # dY, v, k read, grad_q + dY rows written; indices, scores, attn read, dS written
        'sparse_attention_backward_rows': 5 * S * E * 4 + 4 * S * Z * 4,
        # matrix-core path (mfma_attention.hip): every dense operand once + the CSR information.
        # prepare: indices read, the two sets of cell tiles (lower triangle) written
        # (tiles of a lookup pattern are 0 / 1 counts: 128-byte mask form, both orientations)
        'attention_mfma_prepare': S * Z * 4 + 2 * (S // 32) * (S // 32 + 1) // 2 * 128,
        'attention_mfma_forward': 4 * S * E * 4 + S * Z * 4,       # q k v -> y (+ the pattern)
        # q k v dY y read, three gradients written (+ the pattern); two launches
        'attention_mfma_backward': 8 * S * E * 4 + S * Z * 4,
    }
    return per[op] * B


# op (naive_gpt.ext entry) -> the HIP kernel that does its work at this workload, for the
# committed rocprofv3 PMC summary (profiles/*_traffic.json, tools/pmc_traffic.sh)
OP_KERNEL = {
    'sddmm_forward_cuda': 'spt::sddmm_g4_lds_kernel<4>',
    'spmm_forward_cuda': 'spt::spmm_g4_lds_kernel<4, 1>',
    'spmm_transposed': 'spt::spmm_t64_lds_kernel<1>',
    'csr_transpose': 'spt::csr_transpose_bitmap_kernel',
    'lookup_forward_cuda': 'spt::lookup_rows_kernel<1>',
    'softmax_forward_cuda': 'spt::softmax_kernel<16, 0>',
    'softmax_backward_cuda': 'spt::softmax_kernel<16, 1>',
    'softmax_backward_clamped': 'spt::softmax_kernel<16, 2>',
    'pq_encode_heads': 'spt::pq_encode_heads_kernel<8>',
    'cdist_encode': 'spt::cdist_forward_kernel<8>',
    'pq_loss_forward': 'spt::pq_loss_forward_kernel<8>',
    'pq_loss_backward': 'spt::pq_loss_backward_kernel<8>',
    'sparse_attention_forward': 'spt::sparse_attention_forward_kernel<true, true>',
    'sparse_attention_backward_rows': 'spt::sparse_attention_backward_rows_kernel<true, true>',
    'attention_mfma_prepare': 'spt::attention_cell_tiles_kernel<false>',
    'attention_mfma_forward': 'spt::attention_mfma_forward_kernel<true>',
    'attention_mfma_backward': 'spt::attention_mfma_backward_keys_kernel<true, 0>',
}


OP_LAUNCHES = {
    'spmm_transposed': ['spt::permute_values_kernel', 'spt::spmm_t64_lds_kernel<1>'],
    'pq_loss_forward': ['spt::pq_loss_forward_kernel<8>', 'spt::pq_loss_finish_kernel'],
    'pq_loss_backward': ['spt::pq_loss_backward_kernel<8>', 'spt::pq_loss_table_reduce_kernel'],
    'attention_mfma_backward': ['spt::attention_mfma_backward_rows_kernel<true>',
                                'spt::attention_mfma_backward_keys_kernel<true, 0>'],
}


def measured_traffic(op: str):
    """HBM bytes per launch of the op's kernel from the committed PMC passes (FETCH_SIZE
    doubled per the gfx950 note, + WRITE_SIZE), or None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_traffic.json')))
    if not files or op not in OP_KERNEL:
        return None
    table = json.load(open(files[-1]))
    total, found = 0.0, False
    for name in OP_LAUNCHES.get(op, [OP_KERNEL[op]]):
        if name in table:
            total += table[name]['hbm_bytes_per_launch']
            found = True
    return total if found else None


class EventTimer:
    """Brackets every naive_gpt.ext call with HIP events on torch's current stream
    (the stream the kernels are launched on)."""

    OPS = ['cdist_encode', 'pq_encode_heads', 'softmax_backward_clamped',
           'cdist_forward_cuda', 'cdist_backward_cuda',
           'lookup_forward_cuda', 'sddmm_forward_cuda', 'spmm_forward_cuda',
           'spmm_transposed', 'csr_transpose',
           'softmax_forward_cuda', 'softmax_backward_cuda',
           'pq_loss_forward', 'pq_loss_backward', 'sparse_attention_forward',
           'sparse_attention_backward_rows', 'attention_mfma_prepare', 'attention_mfma_forward',
           'attention_mfma_backward']

    def __init__(self):
        from naive_gpt import ext
        self.ext = ext
        self.events = {op: [] for op in self.OPS}
        self.enabled = False
        self.only = None      # restrict the bracketing to one op (the timed region)
        self.pool = []
        self.orig = {op: getattr(ext, op) for op in self.OPS}
        for op in self.OPS:
            setattr(ext, op, self._wrap(op))

    def reserve(self, n_pairs: int):
        """Create the events BEFORE the timed region: hipEventCreate costs ~10 us of host time,
        40 times per step, which the step with the PQ loss (one host sync per step) cannot
        hide behind GPU work."""
        while len(self.pool) < 2 * n_pairs:
            self.pool.append(torch.cuda.Event(enable_timing=True))

    def _wrap(self, op):
        fn = self.orig[op]

        def timed(*args, **kwargs):
            if not self.enabled or (self.only is not None and op != self.only):
                return fn(*args, **kwargs)
            if len(self.pool) < 2:
                self.reserve(64)
            a, b = self.pool.pop(), self.pool.pop()
            a.record()
            out = fn(*args, **kwargs)
            b.record()
            self.events[op].append((a, b))
            return out
        return timed

    def reset(self):
        self.events = {op: [] for op in self.OPS}

    def summary(self):
        out = {}
        for op, pairs in self.events.items():
            if pairs:
                ms = [a.elapsed_time(b) for a, b in pairs]
                out[op] = {'calls': len(ms), 'avg_us': 1e3 * float(np.mean(ms)),
                           'total_ms': float(np.sum(ms))}
        return out


def sparse_step(attn, q, k, v, trigger):
    if trigger:
        attn.arm()        # = trigger.fill_(True) + a host-side note (utils.SparseTuner does the same)
    y = attn(q, k, v, attn_mask=None)
    loss = y.sum()
    if trigger:
        loss = loss + 1e-2 * attn.loss
    loss.backward()


def allreduce_grads(params, world):
    # one flat fp32 buffer, one RCCL all-reduce (naive_gpt/utils/distributed.py).  The only
    # trainable tensors of the attention path are the PQ tables, and they receive a gradient
    # only when the PQ loss is armed: without one there is nothing to exchange and the ranks
    # (identical replicas running the same arming decision) all skip the collective.
    if world == 1 or all(p.grad is None for p in params):
        return
    from naive_gpt import utils
    utils.allreduce_gradients(params, world_size=world)


def timed_loop(fn, steps, warmup, world):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def cpu_baseline(n_seq: int):
    """The oracle's single-core chain on `n_seq` sequences x 16 heads (fwd + bwd)."""
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
    return {'value': n_seq * S / dt, 'unit': 'tokens/s', 'cores': 1, 'kind': 'port',
            'sample': '{} sequences x {} heads x seq {} (fwd+bwd, oracle/spt_oracle.c, '
                      '{:.1f} s)'.format(n_seq, H, S, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # a step is ~0.35 ms: 20 timed steps (7 ms) measured the host's launch jitter as much as the
    # GPU (0.33-0.38 ms/step run to run); 200 steps are still well under a second
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=16, help='micro-batch per GPU')
    ap.add_argument('--trigger', action='store_true', help='arm the PQ training loss')
    ap.add_argument('--no-dense', action='store_true')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--cpu-seqs', type=int, default=40)
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, 'launch {} ranks for --gpus {}'.format(args.gpus, args.gpus)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    from naive_gpt import ext, layers
    ext.load_library()        # fail loudly when the HIP library is missing
    timer = EventTimer()

    torch.manual_seed(rank)
    dev = torch.device('cuda', local_rank)
    N = args.batch
    attn = layers.SparseVanillaAttentionV2(d_head=E, d_codeword=D, n_codewords=C,
                                           p_dropout=0.0).to(dev)
    if world > 1:   # identical replicas
        from naive_gpt import utils
        utils.broadcast_parameters(attn, src=0)
    q, k, v = [torch.randn([N, S, H, E], device=dev, requires_grad=True) for _ in range(3)]
    params = list(attn.parameters())

    def step():
        for t in (q, k, v):
            t.grad = None
        attn.zero_grad(set_to_none=True)
        sparse_step(attn, q, k, v, args.trigger)
        allreduce_grads(params, world)

    torch.cuda.reset_peak_memory_stats()
    # Warm-up (untimed): after the first step every ext op is bracketed with HIP events,
    # which gives the per-op table and names the dominant op.  Bracketing all ~17 launches
    # of a step costs ~0.1 ms of host time per step (measured: 0.87 vs 0.74 ms/step), so in
    # the timed region only the dominant op keeps its events: `value` is not perturbed and
    # the roofline kernel is still measured live inside the K timed steps.
    # (Live HIP events are not free either: with ~600 created-but-unused events every
    # host synchronisation -- one per step when the PQ loss is armed -- took 2 ms longer, so
    # the pool holds exactly what the next loop records and is dropped afterwards.)
    warm_steps = max(args.warmup - 1, 1)
    timed_loop(step, 0, 1, world)
    # A step is < 1 ms: W = 5 warm-up steps end before the GPU has left its idle clocks
    # (observed: the same command 0.74 or 0.97 ms/step).  300 more untimed steps (~0.25 s;
    # a fixed count, so that every rank makes the same number of collective calls).
    timed_loop(step, 0, 300, world)
    timer.reserve(32 * warm_steps)
    timer.enabled = True
    timed_loop(step, 0, warm_steps, world)
    timer.enabled = False
    warm = timer.summary()
    timer.pool.clear()
    dominant = max(warm, key=lambda o: warm[o]['total_ms']) if warm else None
    timer.reset()
    timer.only = dominant
    timer.enabled = dominant is not None
    if dominant is not None:
        timer.reserve(-(-warm[dominant]['calls'] // warm_steps) * args.steps)
    dt = timed_loop(step, args.steps, 0, world)
    timer.enabled = False
    timer.pool.clear()
    peak_gb = torch.cuda.max_memory_allocated() / 1e9
    tokens = N * S * world * args.steps

    # The tuning recipe arms the PQ loss on EVERY step (script/4-sparse-tuning-0.py:71-78);
    # the measurement harness the headline follows (script/0-profile.py) never does.
    # Report the recipe's step next to the headline.
    recipe = None
    if not args.trigger:
        def recipe_step():
            for t in (q, k, v):
                t.grad = None
            attn.zero_grad(set_to_none=True)
            sparse_step(attn, q, k, v, True)
            allreduce_grads(params, world)
        rdt = timed_loop(recipe_step, args.steps, args.warmup, world)
        recipe = {'value': tokens / rdt, 'unit': 'tokens/s', 'ms_per_step': 1e3 * rdt / args.steps,
                  'what': 'same step with the PQ codebook loss armed (kernels.pq_loss fwd+bwd '
                          'for q and k, codebook gradient all-reduced)'}
    result = {
        'metric': 'fine-tune tokens/sec, BERT-large sparse-MHA (fwd+bwd), seq=512',
        'value': tokens / dt, 'unit': 'tokens/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BASELINE.json configs[1]: BERT-large sparse-MHA only '
                               '(cdist/lookup/sddmm/softmax/spmm fwd+bwd)',
                   'micro_batch_per_gpu': N, 'global_batch': N * world, 'seq_len': S,
                   'n_heads': H, 'd_head': E, 'nnz_per_row': Z, 'pq': [M, C, D],
                   'trigger': bool(args.trigger), 'parallelism': 'dp{}'.format(world),
                   'arithmetic': 'fp32 tensors; attention products = 3 bf16 MFMAs on hi/lo-split '
                                 'fp32 operands, fp32 accumulation (error ~2e-5 of the output '
                                 'scale, bar 1e-3); PQ codes / top-k indices exact'},
        'peak_hbm_gb': peak_gb,
    }
    if recipe is not None:
        result['with_pq_loss'] = recipe
    if world > 1:
        # Outside the timed region: the one exchange a real fine-tune step of this model family
        # makes -- the flat fp32 buffer of trainable gradients (~9 M parameters =
        # 36 MB for the 24-layer BERT-large-dims model) -- as RCCL sees it on this node.
        try:
            buf = torch.zeros(9 * 1024 * 1024, dtype=torch.float32, device=q.device)
            for _ in range(3):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            tm = torch.tensor([ms], dtype=torch.float64, device=q.device)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            ms = float(tm.item())
            nbytes = buf.numel() * 4
            result['grad_allreduce'] = {
                'bytes': nbytes, 'ms': ms,
                'busbw_GBps': 2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9,
                'what': 'all-reduce of a 36 MB fp32 trainable-gradient buffer (24-layer model), '
                        'not part of the timed attention step'}
        except Exception as exc:          # never lose the bench line over the extra measurement
            result['grad_allreduce'] = {'error': repr(exc)}

    if rank == 0:
        kernels = warm                                  # every op, from the warm-up steps
        B = N * H
        live = timer.summary()                          # the dominant op, from the timed steps
        kernels[dominant] = live[dominant]
        for op, st in kernels.items():
            st['algorithmic_GBps'] = algorithmic_bytes(op, B) / (st['avg_us'] * 1e-6) / 1e9
            st['per_step'] = st['calls'] / (args.steps if op == dominant else warm_steps)
        st = kernels[dominant]
        result['roofline'] = {
            'kernel': dominant, 'bound': 'hbm', 'achieved': st['algorithmic_GBps'],
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': st['algorithmic_GBps'] / HBM_PEAK_GBS,
            'traffic': measured_traffic(dominant), 'avg_us': st['avg_us'],
            'calls_per_step': st['per_step'],
            'bytes_per_launch': algorithmic_bytes(dominant, B),
            # an ext op can be more than one launch: spmm_transposed = permute_values_kernel
            # + spmm_t64_lds_kernel<1>; the HIP events bracket the op, so avg_us is the SUM
            # of those rows in profiles/*_kernel_stats.csv
            'hip_kernels': OP_LAUNCHES.get(dominant, [OP_KERNEL.get(dominant)]),
        }
        if dominant in MFMA_PER_TILE:
            # the matrix-core kernels are issue-bound, not HBM-bound: also price the MFMAs they
            # execute (dense 32 x 32 tiles on and below the diagonal, three bf16 MFMAs per fp32
            # product) against the dense bf16 peak
            tiles = (S // 32) * (S // 32 + 1) // 2
            flops = B * tiles * MFMA_PER_TILE[dominant] * 2 * 32 * 32 * 16
            tf = flops / (st['avg_us'] * 1e-6) / 1e12
            result['roofline']['mfma'] = {'executed_TFLOPs': tf, 'peak': MFMA_BF16_PEAK_TF,
                                          'frac': tf / MFMA_BF16_PEAK_TF,
                                          'mfma_per_tile': MFMA_PER_TILE[dominant]}
        result['kernels'] = {op: {'avg_us': round(s_['avg_us'], 2),
                                  'calls_per_step': s_['per_step'],
                                  'GBps': round(s_['algorithmic_GBps'], 1),
                                  'frac': round(s_['algorithmic_GBps'] / HBM_PEAK_GBS, 4)}
                             for op, s_ in kernels.items()}

    # dense causal attention on the same GPU: the baseline of the 2x / 50% claims
    if not args.no_dense:
        del attn
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        dense = layers.VanillaAttention(d_head=E, p_dropout=0.0).to(dev)
        mask = torch.full([S, S], float('-inf'), device=dev).triu(1)

        def dense_step():
            for t in (q, k, v):
                t.grad = None
            dense(q, k, v, attn_mask=mask).sum().backward()

        ddt = timed_loop(dense_step, args.steps, args.warmup, world)
        result['dense'] = {'value': tokens / ddt, 'unit': 'tokens/s',
                           'ms_per_step': 1e3 * ddt / args.steps,
                           'peak_hbm_gb': torch.cuda.max_memory_allocated() / 1e9,
                           'what': 'layers.VanillaAttention fwd+bwd, causal mask, fp32'}
        result['speedup_vs_dense'] = result['value'] / result['dense']['value']
        result['peak_hbm_vs_dense'] = peak_gb / result['dense']['peak_hbm_gb']

    if rank == 0 and world == 1 and not args.no_cpu:
        result['cpu_baseline'] = cpu_baseline(args.cpu_seqs)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == '__main__':
    main()
