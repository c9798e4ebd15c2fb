"""naive_gpt.utils (reference: ``naive_gpt/utils/__init__.py``)."""
from .adapter import LoRAHandler
from .adapter import ModuleUpgrader
from .adapter import SparseLoRAHandler
from .distributed import allreduce_gradients
from .distributed import broadcast_parameters
from .distributed import trainable_parameters
from .checkpoint import load_checkpoint
from .checkpoint import model_from_checkpoint
from .checkpoint import save_checkpoint
from .tuning import SparseTuner
from .tuning import upgrade_sparse
from .evaluate import evaluate_mmlu
from .evaluate import load_spt_model
