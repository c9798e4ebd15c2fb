"""MMLU evaluation without Lightning (reference: script/3-mmlu-evaluate.py:12-155).

``load_spt_model``  base checkpoint ({'config', 'state_dict'}, script/1-convert.py:188-195) ->
                    four-stage upgrade -> the tuned adapters' state_dict on top (strict=False; every
                    key the tuned checkpoint lacks must be a frozen base weight, :44-60)
``evaluate_mmlu``   the test loop: batches of ``[position, tokens ..]`` (loaders.TruncPadding),
                    logits of ``batch[:, 1:-1]``, perplexity against ``batch[:, 2:]`` and the answer
                    accuracy with the reference's indexing (:64-93) -- ``SparseTuner.validation_step``
                    is that arithmetic; the means over batches are what Lightning's ``self.log``
                    would have reported.
"""
import math
from typing import Iterable, Optional

import torch

from . import checkpoint
from .tuning import SparseTuner, upgrade_sparse


def load_spt_model(ckpt_path: str, spt_ckpt_path: Optional[str] = None, d_lora: int = 16, device=None):
    model = checkpoint.model_from_checkpoint(ckpt_path)
    model = upgrade_sparse(model, d_lora=d_lora)
    if spt_ckpt_path:
        _, tuned = checkpoint.load_checkpoint(spt_ckpt_path)
        missing = model.load_state_dict(tuned, strict=False).missing_keys
        bad = [name for name in missing if 'lora' in name]
        if bad:
            raise RuntimeError('the tuned checkpoint lacks adapter tables: {}'.format(bad[:4]))
    return model.to(device) if device is not None else model


@torch.no_grad()
def evaluate_mmlu(model, batches: Iterable[torch.Tensor], n_batches: int = 64, device=None) -> dict:
    """Means over at most ``n_batches`` batches (``limit_test_batches`` of the reference's Trainer)."""
    tuner = model if isinstance(model, SparseTuner) else None
    if tuner is None:
        tuner = SparseTuner.__new__(SparseTuner)       # validation_step needs the model and the loss only
        tuner.model = model
        tuner.loss_fn = torch.nn.CrossEntropyLoss()
    device = device if device is not None else next(tuner.model.parameters()).device
    was_training = tuner.model.training
    totals, seen = {'loss': 0.0, 'ppl': 0.0, 'accuracy': 0.0}, 0
    for batch in batches:
        if seen >= n_batches:
            break
        out = tuner.validation_step(batch.to(device))
        for key in totals:
            totals[key] += float(out[key])
        seen += 1
    tuner.model.train(was_training)
    if seen == 0:
        raise RuntimeError('evaluate_mmlu: the loader produced no batch')
    result = {key: value / seen for key, value in totals.items()}
    result['batches'] = seen
    result['ppl_of_mean_loss'] = math.exp(result['loss'])
    return result
