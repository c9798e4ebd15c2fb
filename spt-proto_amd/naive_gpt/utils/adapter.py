"""Module upgrader (reference: ``naive_gpt/utils/adapter.py``).

``ModuleUpgrader(handler).visit(root)`` walks ``root.named_modules()``, asks the
handler's ``on<ClassName>`` method (or ``default``) for a replacement of each module
and swaps the replacements in afterwards.  ``SparseLoRAHandler`` stages the SPT
conversion: ``'lora'`` (every Linear/Embedding -> LoRA), ``'ffn'`` (Feedforward ->
LoRA routed FFN with 4 blocks), ``'mha_v1'`` (attention -> dense attention + PQ
training loss, d_codeword 8, 16 codewords, d_head/8 subspaces), ``'mha_v2'``
(-> PQ sparse attention, codebook carried over).  These constants fix every
hyper-parameter of the hot path.
"""
from torch import nn

from naive_gpt import layers

D_CODEWORD = 8       # reference: adapter.py:97,115
N_CODEWORDS = 16
FFN_BLOCKS = 4       # block_size = d_ff // 4, reference: adapter.py:163,178


def _log(tag: str, name: str, child: nn.Module, new: nn.Module = None):
    if new is None:
        print(tag, name, type(child).__name__)
    else:
        print(tag, name, type(child).__name__, '->', type(new).__name__)


class LoRAHandler:
    def __init__(self, d_lora: int):
        self.d_lora = d_lora

    def default(self, name: str, child: nn.Module):
        _log('[SKIP]', name, child)

    def onLinear(self, name: str, child: nn.Linear):
        assert isinstance(child, nn.Linear)
        new_model = layers.LoRALinear.from_pretrained(d_lora=self.d_lora, source=child)
        _log('[UPGRADE]', name, child, new_model)
        return new_model

    def onEmbedding(self, name: str, child: nn.Embedding):
        assert isinstance(child, nn.Embedding)
        new_model = layers.LoRAEmbedding.from_pretrained(d_lora=self.d_lora, source=child)
        _log('[UPGRADE]', name, child, new_model)
        return new_model


class SparseLoRAHandler(LoRAHandler):
    STAGES = ('lora', 'ffn', 'mha_v1', 'mha_v2')

    def __init__(self, d_lora: int, stage: str):
        super().__init__(d_lora=d_lora)
        assert stage in self.STAGES
        self.stage = stage

    def _inactive(self, stage: str, name: str, child: nn.Module) -> bool:
        if self.stage != stage:
            _log('[SKIP]', name, child)
            return True
        return False

    def onLinear(self, name: str, child: nn.Linear):
        assert isinstance(child, nn.Linear)
        if self._inactive('lora', name, child):
            return None
        return LoRAHandler.onLinear(self, name=name, child=child)

    def onEmbedding(self, name: str, child: nn.Embedding):
        assert isinstance(child, nn.Embedding)
        if self._inactive('lora', name, child):
            return None
        return LoRAHandler.onEmbedding(self, name=name, child=child)

    def _to_v1(self, cls, name: str, child: nn.Module):
        new_model = cls(d_head=child.d_head, p_dropout=child.p_dropout,
                        d_codeword=D_CODEWORD, n_codewords=N_CODEWORDS,
                        n_subspaces=child.d_head // D_CODEWORD)
        _log('[UPGRADE]', name, child, new_model)
        return new_model

    def onVanillaAttention(self, name: str, child: layers.VanillaAttention):
        if self._inactive('mha_v1', name, child):
            return None
        assert isinstance(child, layers.VanillaAttention)
        return self._to_v1(layers.SparseVanillaAttentionV1, name, child)

    def onRotaryAttention(self, name: str, child: layers.RotaryAttention):
        if self._inactive('mha_v1', name, child):
            return None
        assert isinstance(child, layers.RotaryAttention)
        return self._to_v1(layers.SparseRotaryAttentionV1, name, child)

    def _to_v2(self, cls, name: str, child: nn.Module):
        new_model = cls.from_pretrained(source=child)
        _log('[UPGRADE]', name, child, new_model)
        return new_model

    def onSparseVanillaAttentionV1(self, name: str, child: layers.SparseVanillaAttentionV1):
        if self._inactive('mha_v2', name, child):
            return None
        assert isinstance(child, layers.SparseVanillaAttentionV1)
        return self._to_v2(layers.SparseVanillaAttentionV2, name, child)

    def onSparseRotaryAttentionV1(self, name: str, child: layers.SparseRotaryAttentionV1):
        if self._inactive('mha_v2', name, child):
            return None
        assert isinstance(child, layers.SparseRotaryAttentionV1)
        return self._to_v2(layers.SparseRotaryAttentionV2, name, child)

    def _to_routed(self, cls, name: str, child: nn.Module):
        new_model = cls.from_pretrained(
            d_lora=self.d_lora, block_size=child.d_feedforward // FFN_BLOCKS,
            source=child
        )
        _log('[UPGRADE]', name, child, new_model)
        return new_model

    def onFeedforward(self, name: str, child: layers.Feedforward):
        if self._inactive('ffn', name, child):
            return None
        assert isinstance(child, layers.Feedforward)
        return self._to_routed(layers.LoRARoutedFFN, name, child)

    def onLLaMaFeedforward(self, name: str, child: layers.LLaMaFeedforward):
        if self._inactive('ffn', name, child):
            return None
        assert isinstance(child, layers.LLaMaFeedforward)
        return self._to_routed(layers.LoRARoutedLLaMaFFN, name, child)


class ModuleUpgrader:
    def __init__(self, handler: object):
        if not hasattr(handler, 'default'):
            raise RuntimeError('requires default handler')
        self.handler = handler

    def visit(self, root: nn.Module) -> nn.Module:
        # 1. collect replacements; dispatch on the exact class name, so subclasses
        #    are not caught by their parents' hooks (reference: adapter.py:198-203)
        replacements = {}
        for name, child in root.named_modules():
            hook = getattr(self.handler, 'on' + type(child).__name__, None)
            if hook is None:
                hook = self.handler.default
            new_child = hook(name=name, child=child)
            if new_child is None or new_child is child:
                continue
            assert isinstance(new_child, nn.Module)
            replacements[name] = new_child
        # 2. swap them in through their parents
        for path, new_child in replacements.items():
            parent_path, _, leaf = path.rpartition('.')
            parent = root.get_submodule(parent_path) if parent_path else root
            parent.add_module(leaf, module=new_child)
        return root
