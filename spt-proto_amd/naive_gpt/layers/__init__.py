"""naive_gpt.layers -- the class surface of the reference
(``naive_gpt/layers/__init__.py:2-36``): same names, constructor keywords,
``from_pretrained`` statics and state_dict keys."""
# utils
from .basic.utils import FnModule
from .basic.utils import LlamaRMSNorm

# product quantiser
from .basic.quantizer import PQV1
from .basic.quantizer import PQV2

# dense attention
from .basic.position import RotaryEmbedding
from .basic.attention import VanillaAttention
from .basic.attention import RotaryAttention
from .basic.multihead import MultiheadAttention

# dense feed-forward
from .basic.feedforward import Feedforward
from .basic.feedforward import LLaMaFeedforward

# routed feed-forward
from .sparse.feedforward import RoutedFFN
from .sparse.feedforward import RoutedLLaMaFFN

# transformer block
from .basic.transformer import TransformerBlock

# LoRA
from .tuning.lora import LoRALinear
from .tuning.lora import LoRAEmbedding
from .tuning.lora_ffn import LoRARoutedFFN
from .tuning.lora_ffn import LoRARoutedLLaMaFFN

# PQ sparse attention
from .sparse.attention import SparseVanillaAttentionV1
from .sparse.attention import SparseVanillaAttentionV2
from .sparse.attention import SparseRotaryAttentionV1
from .sparse.attention import SparseRotaryAttentionV2
