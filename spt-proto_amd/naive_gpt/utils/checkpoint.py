"""Checkpoint format of the fine-tuning recipe, and the HuggingFace -> naive_gpt weight map.

Format (reference: ``script/1-convert.py:188-195``, consumed by
``script/4-sparse-tuning-0.py:20-31``): ``torch.save({'config': {...}, 'state_dict':
{...}})`` where ``config`` holds exactly the keyword arguments of
``models.OPTModel`` / ``models.LLaMAModel`` and the model family is told from the file
name (``'opt'`` / ``'llama'`` in the path).

Loading uses ``weights_only=True``: a checkpoint is data, nothing in it is executed.

The weight map restates ``script/1-convert.py:8-130`` as two tables (HF parameter name
-> ours); it works on any state_dict with those names, e.g. a randomly initialised
``transformers.OPTForCausalLM(config)`` -- no download is involved here.
"""
import os
from typing import Dict, Tuple

import torch
from torch import nn

from naive_gpt import models

CONFIG_KEYS = ('d_model', 'n_heads', 'n_layers', 'vocab_size', 'd_feedforward',
               'max_length', 'p_dropout')


def model_family(path_or_name: str) -> str:
    name = os.path.basename(str(path_or_name)).lower()
    if 'opt' in name:
        return 'opt'
    if 'llama' in name:
        return 'llama'
    raise RuntimeError('cannot tell the model family (opt / llama) from ' + repr(path_or_name))


def build_model(family: str, config: dict) -> nn.Module:
    missing = [k for k in CONFIG_KEYS if k not in config]
    if missing:
        raise KeyError('checkpoint config lacks ' + ', '.join(missing))
    cls = {'opt': models.OPTModel, 'llama': models.LLaMAModel}[family]
    return cls(**{k: config[k] for k in CONFIG_KEYS})


def save_checkpoint(path: str, config: dict, model: nn.Module) -> None:
    torch.save({'config': dict(config), 'state_dict': model.state_dict()}, f=path)


def load_checkpoint(path: str, map_location='cpu') -> Tuple[dict, Dict[str, torch.Tensor]]:
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    if not isinstance(ckpt, dict) or 'config' not in ckpt or 'state_dict' not in ckpt:
        raise RuntimeError("not a naive_gpt checkpoint: expected {'config', 'state_dict'}")
    return ckpt['config'], ckpt['state_dict']


def model_from_checkpoint(path: str, map_location='cpu') -> nn.Module:
    config, state = load_checkpoint(path, map_location=map_location)
    model = build_model(model_family(path), config)
    model.load_state_dict(state)
    return model


# --------------------------------------------------------------- HuggingFace weight maps

_OPT_LAYER = {
    'fc1': 'ffd.fc1', 'fc2': 'ffd.fc2',
    'self_attn.q_proj': 'mha.linear_q', 'self_attn.k_proj': 'mha.linear_k',
    'self_attn.v_proj': 'mha.linear_v', 'self_attn.out_proj': 'mha.linear_o',
    'self_attn_layer_norm': 'norm1', 'final_layer_norm': 'norm2',
}
_OPT_TOP = {
    'model.decoder.embed_tokens.weight': 'embedding.weight',
    'model.decoder.embed_positions.weight': 'learned_pe.weight',
    'model.decoder.final_layer_norm.weight': 'final_norm.weight',
    'model.decoder.final_layer_norm.bias': 'final_norm.bias',
    'lm_head.weight': 'lm_output.weight',
}
_LLAMA_LAYER = {
    'mlp.gate_proj': 'ffd.gate', 'mlp.up_proj': 'ffd.side', 'mlp.down_proj': 'ffd.down',
    'self_attn.q_proj': 'mha.linear_q', 'self_attn.k_proj': 'mha.linear_k',
    'self_attn.v_proj': 'mha.linear_v', 'self_attn.o_proj': 'mha.linear_o',
    'input_layernorm': 'norm1', 'post_attention_layernorm': 'norm2',
}
_LLAMA_TOP = {
    'model.embed_tokens.weight': 'embedding.weight',
    'model.norm.weight': 'final_norm.weight',
    'lm_head.weight': 'lm_output.weight',
}
_IGNORED_SUFFIXES = ('rotary_emb.inv_freq',)      # recomputed by layers.RotaryEmbedding


def _convert(hf_state: Dict[str, torch.Tensor], layer_prefix: str, layer_map: dict,
             top_map: dict) -> Dict[str, torch.Tensor]:
    out, unknown = {}, []
    for name, tensor in hf_state.items():
        if name.endswith(_IGNORED_SUFFIXES):
            continue
        if name in top_map:
            out[top_map[name]] = tensor
            continue
        if name.startswith(layer_prefix):
            index, _, rest = name[len(layer_prefix):].partition('.')
            module, _, leaf = rest.rpartition('.')
            if index.isdigit() and module in layer_map:
                out['decoders.{}.{}.{}'.format(index, layer_map[module], leaf)] = tensor
                continue
        unknown.append(name)
    if unknown:       # the reference asserts that every tensor was consumed (1-convert.py:69,129)
        raise RuntimeError('unmapped checkpoint tensors: ' + ', '.join(sorted(unknown)[:8]))
    return out


def opt_state_from_hf(hf_state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return _convert(hf_state, 'model.decoder.layers.', _OPT_LAYER, _OPT_TOP)


def llama_state_from_hf(hf_state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return _convert(hf_state, 'model.layers.', _LLAMA_LAYER, _LLAMA_TOP)


def config_from_hf(hf_config) -> dict:
    """``script/1-convert.py:146-160``."""
    d_ff = getattr(hf_config, 'ffn_dim', None) or hf_config.intermediate_size
    return {
        'd_model': hf_config.hidden_size, 'n_heads': hf_config.num_attention_heads,
        'n_layers': hf_config.num_hidden_layers, 'vocab_size': hf_config.vocab_size,
        'd_feedforward': d_ff, 'max_length': hf_config.max_position_embeddings,
        'p_dropout': 0.0,
    }


def load_hf_state(model: nn.Module, family: str, hf_state: Dict[str, torch.Tensor]) -> None:
    """Copy a HuggingFace OPT / LLaMA state_dict into ``model``; buffers that HF does not
    store (``attn_mask``, rotary caches) keep their constructed values."""
    convert = {'opt': opt_state_from_hf, 'llama': llama_state_from_hf}[family]
    report = model.load_state_dict(convert(dict(hf_state)), strict=False)
    if report.unexpected_keys:
        raise RuntimeError('unexpected keys: ' + ', '.join(report.unexpected_keys[:8]))
    params = {n for n, _ in model.named_parameters()}
    lacking = [k for k in report.missing_keys if k in params]
    if lacking:
        raise RuntimeError('parameters without a source tensor: ' + ', '.join(lacking[:8]))
