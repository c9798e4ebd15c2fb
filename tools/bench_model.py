"""BASELINE.json configs[2]: the full fine-tune step at BERT-large dimensions (24 layers,
d_model 1024, 16 heads, d_ff 4096, seq 512, vocab 30522), one MI355X.

Step = utils.SparseTuner.training_step (script/4-sparse-tuning-0.py restated: PQ triggers
armed, CE + 1e-2 aux, backward, clip 1.0, AdamW) on synthetic tokens.  Three tunings of the
same random-init model, protocol of script/0-profile.py:203-226:
  full   -- dense attention + dense FFN, every parameter trained
  lora   -- dense model, LoRA adapters only
  sparse -- SPT: LoRA + routed FFN + PQ sparse attention (the four-stage upgrade)
Prints one JSON object; tokens/s and peak HBM per tuning plus the sparse/full ratios."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import models, utils

# other configurations by environment, e.g. BASELINE configs[4] (LLaMA-7B dims):
#   FAMILY=llama D_MODEL=4096 N_HEADS=32 D_FF=11008 LAYERS=32 SEQ=2048 VOCAB=32000 BATCH=1
FAMILY = os.environ.get('FAMILY', 'opt')
S = int(os.environ.get('SEQ', 512))
CONFIG = dict(d_model=int(os.environ.get('D_MODEL', 1024)), n_heads=int(os.environ.get('N_HEADS', 16)),
              n_layers=int(os.environ.get('LAYERS', 24)), max_length=S,
              vocab_size=int(os.environ.get('VOCAB', 30522)),
              d_feedforward=int(os.environ.get('D_FF', 4096)), p_dropout=0.0)
N = int(os.environ.get('BATCH', 16))
dev = 'cuda'


def build(tuning):
    torch.manual_seed(0)
    with torch.device(dev):           # (a 7B model is built on the GPU, not copied to it)
        model = (models.LLaMAModel if FAMILY == 'llama' else models.OPTModel)(**CONFIG)
        if tuning == 'lora':
            model = utils.upgrade_sparse(model, d_lora=16, stages=('lora',))
        elif tuning == 'sparse':
            model = utils.upgrade_sparse(model, d_lora=16)
    return model.to(dev)


def run(tuning, steps=int(os.environ.get('STEPS', 8)), warmup=3):
    model = build(tuning)
    tuner = utils.SparseTuner(model)
    gen = torch.Generator(device=dev).manual_seed(1)

    def step():
        batch = torch.randint(3, CONFIG['vocab_size'], [N, S + 2], device=dev, generator=gen)
        tuner.training_step(batch, pq_loss=(tuning == 'sparse'))

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
           'trainable_params': sum(p.numel() for p in tuner.params),
           'total_params': sum(p.numel() for p in model.parameters())}
    del model, tuner
    torch.cuda.empty_cache()
    return res


out = {'config': dict(CONFIG, family=FAMILY, batch=N, seq=S, dtype='f32',
                      what='SparseTuner.training_step: fwd + bwd + clip + AdamW')}
tunings = os.environ.get('TUNINGS', 'full,lora,sparse').split(',')
for tuning in tunings:
    out[tuning] = run(tuning)
    print(tuning, out[tuning], file=sys.stderr, flush=True)
if 'sparse' in out and 'full' in out:
    out['sparse_vs_full_speedup'] = out['sparse']['tokens_per_s'] / out['full']['tokens_per_s']
    out['sparse_vs_full_peak_mem'] = out['sparse']['peak_hbm_gb'] / out['full']['peak_hbm_gb']
if 'sparse' in out and 'lora' in out:
    out['sparse_vs_lora_speedup'] = out['sparse']['tokens_per_s'] / out['lora']['tokens_per_s']
    out['sparse_vs_lora_peak_mem'] = out['sparse']['peak_hbm_gb'] / out['lora']['peak_hbm_gb']
print(json.dumps(out))
