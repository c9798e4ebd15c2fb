"""One TransformerBlock at BERT-large dimensions (the reference's `opt-1024`), protocol of
script/0-profile.py:151-226: dense full fine-tune vs LoRA vs SPT sparse (LoRA + routed FFN +
PQ sparse attention); fwd + bwd + AdamW step, tokens/s and peak HBM."""
import contextlib, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from torch import nn, optim
from naive_gpt import layers, utils

# FAMILY=llama: RMSNorm, rotary attention, SiLU-gated FFN (BASELINE config 5 with
# D_MODEL=4096 N_HEADS=32 D_FF=11008 SEQ=2048 BATCH=1)
family = os.environ.get('FAMILY', 'opt')
d_model, n_heads, d_ff = (int(os.environ.get('D_MODEL', 1024)), int(os.environ.get('N_HEADS', 16)),
                          int(os.environ.get('D_FF', 4096)))
N, S = int(os.environ.get('BATCH', 16)), int(os.environ.get('SEQ', 512))
dev = 'cuda'


def build(tuning):
    torch.manual_seed(0)
    if family == 'llama':
        model = layers.TransformerBlock(
            d_model=d_model, n_heads=n_heads, layernorm_fn=layers.LlamaRMSNorm(d_model),
            attention_fn=layers.RotaryAttention(d_head=d_model // n_heads, p_dropout=0.0,
                                                max_length=S),
            feedforward_fn=layers.LLaMaFeedforward(d_model=d_model, d_feedforward=d_ff,
                                                   activation=nn.SiLU()),
            attention_bias=False, pre_norm=True)
    else:
        model = layers.TransformerBlock(
            d_model=d_model, n_heads=n_heads, layernorm_fn=nn.LayerNorm(d_model),
            attention_fn=layers.VanillaAttention(d_head=d_model // n_heads, p_dropout=0.0),
            feedforward_fn=layers.Feedforward(d_model=d_model, d_feedforward=d_ff,
                                              activation=nn.ReLU(), p_dropout=0.0),
            attention_bias=True, pre_norm=True)
    with contextlib.redirect_stdout(io.StringIO()):
        if tuning == 'lora':
            model = utils.ModuleUpgrader(utils.LoRAHandler(d_lora=16)).visit(model)
        elif tuning == 'sparse':
            for stage in ['lora', 'ffn', 'mha_v1', 'mha_v2']:
                model = utils.ModuleUpgrader(utils.SparseLoRAHandler(d_lora=16, stage=stage)).visit(model)
    return model.to(dev)


def run(tuning, steps=10, warmup=5):
    model = build(tuning)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = optim.AdamW(params, lr=1e-4, weight_decay=1e-2, fused=True)
    mask = None if tuning == 'sparse' else torch.full([S, S], float('-inf'), device=dev).triu(1)

    def step():
        x = torch.randn([N, S, d_model], device=dev, requires_grad=True)
        y = model(x, attn_mask=mask)
        y.sum().backward()
        opt.step()
        model.zero_grad()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = {'ms_per_step': dt * 1e3, 'tokens_per_s': N * S / dt,
           'peak_hbm_gb': torch.cuda.max_memory_allocated() / 1e9,
           'trainable_params': sum(p.numel() for p in params)}
    del model, opt
    torch.cuda.empty_cache()
    return res


out = {'config': {'family': family, 'd_model': d_model, 'n_heads': n_heads, 'd_ff': d_ff, 'batch': N, 'seq': S,
                  'dtype': 'f32', 'what': 'one TransformerBlock, fwd+bwd+AdamW'}}
tunings = os.environ.get('TUNINGS', 'full,lora,sparse').split(',')
for tuning in tunings:
    out[tuning] = run(tuning)
if len(tunings) < 3:
    print(json.dumps(out))
    sys.exit(0)
out['sparse_vs_full_speedup'] = out['sparse']['tokens_per_s'] / out['full']['tokens_per_s']
out['sparse_vs_lora_speedup'] = out['sparse']['tokens_per_s'] / out['lora']['tokens_per_s']
out['sparse_vs_full_peak_mem'] = out['sparse']['peak_hbm_gb'] / out['full']['peak_hbm_gb']
out['sparse_vs_lora_peak_mem'] = out['sparse']['peak_hbm_gb'] / out['lora']['peak_hbm_gb']
print(json.dumps(out))
