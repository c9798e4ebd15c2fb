"""SPT fine-tuning step without Lightning.

Restates ``script/4-sparse-tuning-0.py``: ``LightningModel`` (:12-126) and the
``L.Trainer`` settings it runs under (:176-187) --

* model from a ``{'config', 'state_dict'}`` checkpoint, upgraded in the four stages
  ``lora -> ffn -> mha_v1 -> mha_v2`` with ``SparseLoRAHandler`` (:33-39);
* AdamW(lr 1e-4, weight_decay 1e-1) + ExponentialLR(gamma 0.9) stepped per epoch (:45-53);
* one step: arm every ``*.trigger`` buffer (:71-78), next-token cross entropy on
  ``batch[:, 1:-1] -> batch[:, 2:]`` (column 0 carries the MMLU answer position)
  (:79-81), plus ``1e-2 *`` the sum of the ``*.loss`` buffers the sparse attentions
  registered (:83-90);
* ``accumulate_grad_batches``, ``gradient_clip_val=1.0`` (L2 norm), fp32 (:183-187);
* DDP over the GPUs of the node (:183) -- here: identical replicas, gradients averaged
  with one flat RCCL all-reduce (``utils.distributed``) before clipping, so every rank
  clips by the same norm and takes the same step.

No dataset code: ``training_step`` takes the token batch the reference's data module
would deliver.
"""
import contextlib
import io
import math
import weakref
from typing import Iterable, Optional

import torch
from torch import nn, optim

from . import adapter, checkpoint, distributed

STAGES = ('lora', 'ffn', 'mha_v1', 'mha_v2')


def upgrade_sparse(model: nn.Module, d_lora: int, stages: Iterable[str] = STAGES,
                   verbose: bool = False) -> nn.Module:
    """The four-stage upgrade of script/4-sparse-tuning-0.py:33-39 (the handler prints one
    line per replaced module; silenced unless ``verbose``)."""
    sink = contextlib.nullcontext() if verbose else contextlib.redirect_stdout(io.StringIO())
    with sink:
        for stage in stages:
            handler = adapter.SparseLoRAHandler(d_lora=d_lora, stage=stage)
            model = adapter.ModuleUpgrader(handler=handler).visit(model)
    return model


class SparseTuner:
    FLAT_LIMIT = 1 << 26        # trainable elements up to which they are kept in one flat buffer

    def __init__(self, model: nn.Module, lr: float = 1e-4, weight_decay: float = 1e-1,
                 gamma: float = 0.9, clip_norm: Optional[float] = 1.0,
                 aux_weight: float = 1e-2, n_accumulate: int = 1, group=None):
        self.model = model
        self.group = group
        self.world_size = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world_size = torch.distributed.get_world_size(group)
        if self.world_size > 1:
            distributed.broadcast_parameters(model, src=0, group=group)
        self.params = distributed.trainable_parameters(model)
        # torch's fused AdamW on the GPU: the whole update is ONE multi-tensor kernel (the
        # default "foreach" form is ~66 launches and 1.06 ms of the 63 ms BERT-large step; with
        # `capturable=True` it falls to per-parameter divisions, 1,040 launches and 3.8 ms)
        self._fused = bool(self.params) and all(p.is_cuda for p in self.params)
        # The trainable parameters (464 small tensors in the 24-layer model) live in ONE flat
        # buffer -- each `p.data` a view of it -- and so do their gradients once a backward has
        # produced them: the exchange (all-reduce), the clip and AdamW are then a handful of
        # launches on one tensor instead of ~40 multi-tensor launches at N = 1 and 928 copies
        # around the all-reduce at N > 1.  Same arithmetic element by element as long as every
        # parameter receives a gradient in every step (the recipe arms every trigger every step);
        # the first step in which one does not, the tuner goes back to per-parameter entries for
        # good (`_unflatten`: torch's AdamW skips such a parameter).  (Up to
        # FLAT_LIMIT elements: the gathered gradient is a second copy of every gradient while the
        # update runs -- nothing for adapter tables, 1.5 GB for a dense 365 M-parameter fine-tune,
        # which keeps the per-tensor path.)
        self._flat = self._flat_grad = None
        if self.params and len({(p.device, p.dtype) for p in self.params}) == 1 \
                and self.params[0].dtype == torch.float32 \
                and sum(p.numel() for p in self.params) <= self.FLAT_LIMIT:
            self._flatten()
        self.optimizer = optim.AdamW(self._optimised(), lr=lr, weight_decay=weight_decay,
                                     **({'fused': True} if self._fused else {}))
        self.scheduler = optim.lr_scheduler.ExponentialLR(self.optimizer, gamma=gamma)
        self.loss_fn = nn.CrossEntropyLoss()
        self.clip_norm = clip_norm
        self.aux_weight = aux_weight
        self.n_accumulate = max(1, int(n_accumulate))
        self._micro = 0
        self._triggers = [b for n, b in model.named_buffers() if n.endswith('.trigger')]
        # layers that can be armed from the host without a device read-back (layers.sparse)
        self._armable = [m for m in model.modules()
                         if hasattr(m, 'arm') and isinstance(getattr(m, 'trigger', None), torch.Tensor)]
        self._submodules = None
        self.last_grad_norm = None
        self._graph = None
        self._graph_lr = None
        self._graph_launches = -1          # ext.LAUNCHES behind the last replay
        self._graph_keepalive = None       # what the captured kernels read out of module-level caches
        # measurement hook (bench.py at N > 1): HIP events around the gradient exchange of every
        # `allreduce_every`-th update; `allreduce_events` = [(start, end), ...]
        self.allreduce_every = 0
        self.allreduce_events = []
        self._updates = 0

    @classmethod
    def from_checkpoint(cls, path: str, d_lora: int = 16, device=None, **kwargs):
        model = checkpoint.model_from_checkpoint(path)
        model = upgrade_sparse(model, d_lora=d_lora)
        if device is not None:
            model = model.to(device)
        return cls(model, **kwargs)

    # ------------------------------------------------------------------ pieces of a step
    def arm_triggers(self) -> None:
        armed = set()
        for module in self._armable:
            module.arm()
            armed.add(module.trigger.data_ptr())
        for trigger in self._triggers:            # any other module with a `trigger` buffer
            if trigger.data_ptr() not in armed:
                trigger.fill_(True)

    def aux_loss(self):
        """Sum of the PQ codebook losses the armed attentions left in ``*.loss``."""
        # (every sub-module's `loss` buffer, as `named_buffers()` names ending in '.loss' -- from a
        # module list made once: the walk over a 24-layer model's ~930 modules is ~1 ms of host
        # time per step, and the step is within 10 % of being host-bound)
        # (the list is re-made when any module's set of children has changed -- an upgrade or a
        # replaced layer after the first step: ~0.1 ms for the fingerprint against ~1 ms for the walk)
        mark = None
        if self._submodules is not None:
            mark = hash(tuple(id(c) for m in self._submodules_all for c in m._modules.values()))
        if self._submodules is None or mark != self._submodules_mark:
            self._submodules_all = list(self.model.modules())
            self._submodules = [m for m in self._submodules_all if m is not self.model]
            self._submodules_mark = hash(tuple(id(c) for m in self._submodules_all
                                               for c in m._modules.values()))
        self._loss_modules = [m for m in self._submodules if m._buffers.get('loss') is not None]
        losses = [m._buffers['loss'] for m in self._loss_modules]
        if not losses:
            return 0.0
        if len(losses) == 1 or any(l.dim() != 0 for l in losses):
            return sum(losses[1:], losses[0])
        return torch.stack(losses).sum()          # one launch forward, one backward

    def shared_step(self, src: torch.Tensor, target: torch.Tensor):
        output = self.model(src)
        loss = self.loss_fn(output.flatten(end_dim=-2), target=target.flatten())
        return output, loss

    def step_loss(self, src: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """`shared_step(...)[-1]` without materialising what only the loss needs: a frozen LoRA
        head and `nn.CrossEntropyLoss()` go through one fused function
        (layers/tuning/head_loss.py: logits written once, turned into their gradient in place)."""
        from naive_gpt.layers.tuning import head_loss
        model, fn = self.model, self.loss_fn
        plain = (type(fn) is nn.CrossEntropyLoss and fn.weight is None and fn.reduction == 'mean'
                 and fn.label_smoothing == 0.0)
        # (`model.hidden` / `model.lm_output` are called directly: a model with hooks of its own
        # goes through `model(...)` so that they fire)
        hooked = bool(model._forward_hooks or model._forward_pre_hooks
                      or getattr(getattr(model, 'lm_output', None), '_forward_hooks', None))
        if plain and not hooked and hasattr(model, 'hidden') and hasattr(model, 'lm_output') \
                and torch.is_grad_enabled():
            from naive_gpt import ext
            h = model.hidden(src)
            if head_loss.fused_usable(model.lm_output, h):
                ext.note_path('head_loss', 'fused')
                return head_loss.lm_head_loss(model.lm_output, h, target, fn.ignore_index)
            ext.note_path('head_loss', 'library', fallback=h.is_cuda,
                          why=lambda: 'head {} on h {}: the fused loss takes a frozen bias-free LoRALinear, '
                              'fp32, >= 2048 tokens'.format(type(model.lm_output).__name__, tuple(h.shape)))
            return fn(model.lm_output(h).flatten(end_dim=-2), target=target.flatten())
        return self.shared_step(src, target)[-1]

    def training_step(self, batch: torch.Tensor, pq_loss: bool = True) -> torch.Tensor:
        """One micro-batch: forward, backward; every ``n_accumulate``-th call also
        exchanges, clips and applies the gradients.  Returns the (detached) loss.
        After `capture()` a batch of the captured shape replays the HIP graph (the returned loss is
        a copy of the graph's static output: callers may keep it across steps)."""
        if self._graph is not None and batch.shape == self._graph_batch.shape \
                and pq_loss == self._graph_pq:
            if not self.model.training:
                self.model.train()
            from naive_gpt import ext
            if ext.LAUNCHES != self._graph_launches:
                # Eager launches of this library (this model's other batch shapes or validation, or
                # any other model's step) reached the stream since the last replay.  Round 3 saw a
                # replay in that position diverge from its eager twin on ROCm 7.2 -- for one
                # allocation pattern, never behind a host synchronisation -- so the
                # stream is drained first.  Replays back to back, a training loop's case, never wait.
                torch.cuda.current_stream(self._graph_batch.device).synchronize()
            # the batch enters, and the loss leaves, the graph's static tensors through elementwise
            # KERNELS: a same-dtype copy_ / clone is a hipMemcpyAsync, the class of node (memset /
            # memcpy beside kernel nodes) that 5.13 item 3 showed to lose its place
            torch.add(batch, 0, out=self._graph_batch)
            self._graph.replay()
            loss = torch.add(self._graph_loss, 0)
            self._graph_launches = ext.LAUNCHES
            return loss
        return self._eager_step(batch, pq_loss)

    def _eager_step(self, batch: torch.Tensor, pq_loss: bool = True) -> torch.Tensor:
        assert batch.dim() == 2
        if not self.model.training:
            self.model.train()
        if pq_loss:
            self.arm_triggers()
        loss = self.step_loss(batch[:, 1:-1], target=batch[:, 2:])
        if pq_loss:
            loss = loss + self.aux_weight * self.aux_loss()
        (loss / self.n_accumulate).backward()
        if pq_loss:
            # The `loss` buffers are graph outputs: left as they are (the reference does) each keeps
            # its step's autograd graph alive until the layer's next forward -- the attention nodes
            # with their cell tiles (17 MB a layer), and every AccumulateGrad node, whose stream is
            # then the PREVIOUS step's: under `capture()` that pulls the warm-up stream into the
            # capture (torch warns: "may break CUDA graph capture"; it did, round 3).  Keep the values.
            for m in getattr(self, '_loss_modules', ()):
                held = m._buffers.get('loss')
                if held is not None and held.grad_fn is not None:
                    m._buffers['loss'] = held.detach()
        self._micro += 1
        if self._micro % self.n_accumulate == 0:
            self.apply_gradients()
        return loss.detach()

    FLAT_ALIGN = 4          # elements: every slice starts on a 16-byte boundary (float4 kernels read
                            # gamma / beta, LoRA tables and biases straight out of these views)

    def _flatten(self) -> None:
        for p in self.params:
            owner = getattr(p, '_spt_flat_owner', None)
            owner = owner() if owner is not None else None
            if owner is not None and owner is not self and owner._flat is not None:
                raise RuntimeError(
                    'SparseTuner: these parameters already live in the flat buffer of another live '
                    'SparseTuner (its optimiser would go on updating memory the model no longer '
                    'reads): call `release()` on that tuner first')
        sizes = [p.numel() for p in self.params]
        align = self.FLAT_ALIGN
        offsets, total = [], 0
        for n in sizes:
            offsets.append(total)
            total += (n + align - 1) // align * align
        first = self.params[0]
        with torch.no_grad():
            flat = torch.zeros([total], dtype=first.dtype, device=first.device)
            for p, n, offset in zip(self.params, sizes, offsets):
                view = flat[offset:offset + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p._spt_flat_owner = weakref.ref(self)
        self._flat = nn.Parameter(flat)
        self._flat_grad = torch.zeros_like(flat)
        self._sizes, self._offsets = sizes, offsets
        # the gaps behind slices whose length is not a multiple of FLAT_ALIGN: zero gradient there
        # (zero parameter, zero moments: AdamW leaves them at zero)
        self._pads = {}
        for i, (n, offset) in enumerate(zip(sizes, offsets)):
            end = offsets[i + 1] if i + 1 < len(offsets) else total
            if end - offset != n:
                self._pads[i] = torch.zeros([end - offset - n], dtype=first.dtype, device=first.device)

    def _check_flat_views(self) -> None:
        """Every trainable `p.data` must still be ITS slice of the flat buffer (the optimiser only
        owns the buffer).  `model.to()`, `.float()`, `p.data = ...` or a checkpoint loaded with
        `assign=True` re-point a parameter silently: its current values are then copied back into
        the slice and the view restored; another dtype or device cannot be expressed and raises."""
        base, item = self._flat.data_ptr(), self._flat.element_size()
        for p, n, offset in zip(self.params, self._sizes, self._offsets):
            if p.data_ptr() == base + offset * item:
                continue
            if p.dtype != self._flat.dtype or p.device != self._flat.device or p.numel() != n:
                raise RuntimeError(
                    'SparseTuner: a trainable parameter was moved to {} / {} after the tuner was '
                    'built; build the tuner after `model.to(...)`'.format(p.device, p.dtype))
            with torch.no_grad():
                view = self._flat.data[offset:offset + n].view(p.shape)
                view.copy_(p.data)
                p.data = view

    def release(self) -> None:
        """Give the parameters back: one optimiser entry per parameter from now on (state carried
        over), no flat buffer; another tuner may then be built on the same model."""
        if self._flat is not None:
            self._unflatten()
        for p in self.params:
            if getattr(p, '_spt_flat_owner', None) is not None and p._spt_flat_owner() is self:
                p._spt_flat_owner = None

    def _optimised(self):
        return [self._flat] if self._flat is not None else self.params

    def _gather_gradients(self) -> None:
        """p.grad of every trainable parameter, side by side in `_flat_grad` (torch.cat: one
        launch per 128 tensors)."""
        if not self._pads:
            parts = [p.grad.reshape(-1) for p in self.params]
        else:
            parts = []
            for i, p in enumerate(self.params):
                parts.append(p.grad.reshape(-1))
                if i in self._pads:
                    parts.append(self._pads[i])
        torch.cat(parts, out=self._flat_grad)
        self._flat.grad = self._flat_grad

    def _unflatten(self) -> None:
        """Back to one optimiser entry per parameter, for good, carrying the state over: torch's
        AdamW SKIPS a parameter that has no gradient (no decay, no moment update, its own step
        count), which one flat tensor cannot express.  Until now every parameter took every step,
        so the flat state's slices and its step count ARE the per-parameter state."""
        old, group = self.optimizer, self.optimizer.param_groups[0]
        new = optim.AdamW(self.params, lr=group['lr'], betas=group['betas'], eps=group['eps'],
                          weight_decay=group['weight_decay'], capturable=group.get('capturable', False),
                          **({'fused': True} if group.get('fused') else {}))
        if 'initial_lr' in group:
            new.param_groups[0]['initial_lr'] = group['initial_lr']
        state = old.state.get(self._flat)
        if state:
            for p, n, offset in zip(self.params, self._sizes, self._offsets):
                new.state[p] = {
                    'step': state['step'].clone() if torch.is_tensor(state['step']) else state['step'],
                    'exp_avg': state['exp_avg'][offset:offset + n].view_as(p).clone(),
                    'exp_avg_sq': state['exp_avg_sq'][offset:offset + n].view_as(p).clone()}
        self.optimizer = new
        if isinstance(self.scheduler.optimizer, optim.AdamW):       # (not the capturable tuner's stand-in)
            self.scheduler.optimizer = new
        self._flat = self._flat_grad = None     # (the parameters stay views of the buffer: harmless)

    @contextlib.contextmanager
    def _exchange_timer(self):
        """Events on the CURRENT stream around the exchange: the collective runs on the backend's
        own stream, which waits for the current one and which the current one then waits for
        (synchronous all_reduce), so the pair brackets it."""
        self._updates += 1
        timed = (self.allreduce_every > 0 and self._updates % self.allreduce_every == 0
                 and self.params[0].is_cuda)
        if timed:
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
        yield
        if timed:
            end.record()
            self.allreduce_events.append((start, end))

    def apply_gradients(self) -> None:
        if self._flat is not None:
            self._check_flat_views()
        if self._flat is not None and any(p.grad is None for p in self.params):
            self._unflatten()
        if self._flat is None:
            if self.world_size > 1:
                with self._exchange_timer():
                    distributed.allreduce_gradients(self.params, group=self.group,
                                                    world_size=self.world_size)
            if self.clip_norm is not None:
                self.last_grad_norm = nn.utils.clip_grad_norm_(self.params, self.clip_norm)
            self.optimizer.step()
            self.optimizer.zero_grad(set_to_none=True)
            return
        with torch.no_grad():
            self._gather_gradients()
            grad = self._flat_grad
            if self.world_size > 1:             # the whole exchange: one all-reduce, in place
                with self._exchange_timer():
                    torch.distributed.all_reduce(grad, op=torch.distributed.ReduceOp.SUM, group=self.group)
                grad.div_(self.world_size)
            if self.clip_norm is not None:
                # nn.utils.clip_grad_norm_ (script/4-sparse-tuning-0.py: gradient_clip_val): the
                # 2-norm over all gradients, coefficient max_norm / (norm + 1e-6) capped at 1
                norm = torch.linalg.vector_norm(grad)
                grad.mul_(torch.clamp(self.clip_norm / (norm + 1e-6), max=1.0))
                self.last_grad_norm = norm
        self.optimizer.step()
        for p in self.params:
            p.grad = None

    def end_epoch(self) -> None:
        self.scheduler.step()
        if self._graph_lr is not None:
            # the capturable AdamW reads its learning rate from a device tensor
            self._graph_lr.fill_(self.scheduler.get_last_lr()[0])

    # ------------------------------------------------------------------ one step as a HIP graph
    def use_capturable_optimizer(self) -> None:
        """AdamW with `capturable=True` and its learning rate in a device tensor (what a captured
        step needs; the same arithmetic up to rounding).  The optimiser state starts afresh."""
        device = self.params[0].device
        lr = self.lr
        self._graph_lr = torch.tensor(lr, dtype=torch.float32, device=device)
        group = self.optimizer.param_groups[0]
        self.optimizer = optim.AdamW(self._optimised(), lr=self._graph_lr, betas=group['betas'],
                                     eps=group['eps'], weight_decay=group['weight_decay'],
                                     capturable=True, **({'fused': True} if self._fused else {}))
        gamma = self.scheduler.gamma
        self.scheduler = optim.lr_scheduler.ExponentialLR(
            optim.SGD([torch.zeros(1, requires_grad=True)], lr=lr), gamma=gamma)

    def capture(self, batch_shape, pq_loss: bool = True, warmup: int = 3, example=None) -> None:
        """Capture `training_step` (forward, backward, clip, AdamW) for token batches of
        `batch_shape` as ONE HIP graph (torch.cuda.CUDAGraph): a step is ~3,900 kernel launches
        at BERT-large dimensions, and a replay costs the host one call.  Every launch of
        libspt_hip goes to torch's current stream with caller-allocated outputs, so the
        kernels capture as they are; what changes is the optimiser (AdamW `capturable=True`,
        learning rate in a device tensor) -- same arithmetic.  Single process only: the
        data-parallel step keeps the eager path (its all-reduce is not captured here).
        Afterwards `training_step` replays the graph for batches of that shape.
        The `warmup` steps before the capture are real optimisation steps, on `example` (a
        token batch of that shape) or on an all-zero batch; the optimiser state starts afresh."""
        if self.world_size != 1:
            raise RuntimeError('SparseTuner.capture: single-process only')
        if self.n_accumulate != 1:
            raise RuntimeError('SparseTuner.capture: n_accumulate == 1 only')
        device = self.params[0].device
        self.use_capturable_optimizer()
        self._graph_batch = torch.zeros(batch_shape, dtype=torch.long, device=device)
        if example is not None:
            self._graph_batch.copy_(example)
        self._graph_pq = pq_loss
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):       # warm-up off the capture: lazy initialisations
            for _ in range(warmup):
                self._eager_step(self._graph_batch, pq_loss)
        torch.cuda.current_stream(device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._graph_loss = self._eager_step(self._graph_batch, pq_loss)
        self._graph = graph
        # the captured kernels hold ADDRESSES: whatever they read out of the package's caches (the
        # frozen weights' row norms of the ReLU bound, kept weight images) must outlive the cache
        # entry -- `ext.drop_images()` under a live graph was a use-after-free waiting to happen
        from naive_gpt import ext
        self._graph_keepalive = ext.held_by_caches()
        self._graph_launches = -1

    @torch.no_grad()
    def validation_step(self, batch: torch.Tensor) -> dict:
        """Perplexity and the MMLU answer accuracy of script/4-sparse-tuning-0.py:95-126.
        The accuracy keeps the reference's indexing: ``batch[:, position]`` with a vector
        ``position`` selects a [B, B] grid (every row against every row's answer
        position), and the mean runs over that grid."""
        assert batch.dim() == 2
        self.model.eval()
        target = batch[:, 2:]
        output, loss = self.shared_step(batch[:, 1:-1], target=target)
        position = batch[:, 0]
        answers = batch[:, position]
        predict = torch.argmax(output[:, position - 2, :], dim=-1)
        accuracy = torch.eq(predict, answers).float().mean()
        return {'loss': loss, 'ppl': torch.exp(loss), 'accuracy': accuracy}

    @property
    def lr(self) -> float:
        return float(self.optimizer.param_groups[0]['lr'])
