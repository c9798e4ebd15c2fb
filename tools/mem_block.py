"""What one upgraded TransformerBlock keeps alive between forward and backward: every tensor autograd
saves (unique storages, by size) plus allocator marks around the step.  Same model as bench_block.py."""
import contextlib, io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from torch import nn, optim
from naive_gpt import layers, utils

d_model, n_heads, d_ff, N, S = 1024, 16, 4096, 16, 512
dev = 'cuda'
torch.manual_seed(0)
model = layers.TransformerBlock(
    d_model=d_model, n_heads=n_heads, layernorm_fn=nn.LayerNorm(d_model),
    attention_fn=layers.VanillaAttention(d_head=d_model // n_heads, p_dropout=0.0),
    feedforward_fn=layers.Feedforward(d_model=d_model, d_feedforward=d_ff,
                                      activation=nn.ReLU(), p_dropout=0.0),
    attention_bias=True, pre_norm=True)
with contextlib.redirect_stdout(io.StringIO()):
    for stage in ['lora', 'ffn', 'mha_v1', 'mha_v2']:
        model = utils.ModuleUpgrader(utils.SparseLoRAHandler(d_lora=16, stage=stage)).visit(model)
model = model.to(dev)
params = [p for p in model.parameters() if p.requires_grad]
opt = optim.AdamW(params, lr=1e-4, weight_decay=1e-2, fused=True)
param_ptrs = {p.untyped_storage().data_ptr() for p in model.parameters()}
param_ptrs |= {b.untyped_storage().data_ptr() for b in model.buffers()}


def step(hooks=None):
    x = torch.randn([N, S, d_model], device=dev, requires_grad=True)
    with (hooks or contextlib.nullcontext()):
        y = model(x, attn_mask=None)
    marks['after_forward'] = torch.cuda.memory_allocated() / 1e6
    y.sum().backward()
    opt.step()
    model.zero_grad()


marks = {}
for _ in range(3):
    step()
torch.cuda.synchronize()
marks['resident'] = torch.cuda.memory_allocated() / 1e6
import gc
gc.collect()
marks['resident_after_gc'] = torch.cuda.memory_allocated() / 1e6
live = {}
for o in gc.get_objects():
    if isinstance(o, torch.Tensor) and o.is_cuda:
        st = o.untyped_storage()
        if st.data_ptr() not in param_ptrs:
            live[st.data_ptr()] = (st.nbytes(), tuple(o.shape), str(o.dtype))
marks['live_non_param'] = sorted(live.values(), reverse=True)[:12]
torch.cuda.reset_peak_memory_stats()
seen = {}


def pack(t):
    if t.is_cuda:
        st = t.untyped_storage()
        if st.data_ptr() not in param_ptrs:
            seen.setdefault(st.data_ptr(), (st.nbytes(), tuple(t.shape), str(t.dtype)))
    return t


step(torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t))
torch.cuda.synchronize()
marks['peak'] = torch.cuda.max_memory_allocated() / 1e6
rows = sorted(seen.values(), reverse=True)
print(json.dumps({'marks_mb': marks, 'saved_mb': sum(r[0] for r in rows) / 1e6,
                  'saved': [[round(r[0] / 1e6, 2), list(r[1]), r[2]] for r in rows[:40]]}, indent=1))
