"""naive_gpt.kernels -- the five operators of the SPT hot path
(reference: ``naive_gpt/kernels/__init__.py:2-14``)."""
from .cdist import cdist
from .lookup import lookup
from .softmax import softmax
from .sddmm import sddmm
from .spmm import spmm
from .pq_loss import pq_loss      # MI355X addition: fused PQ training loss
from .pq_loss import pq_loss_and_codes

__all__ = ['cdist', 'lookup', 'softmax', 'sddmm', 'spmm', 'pq_loss', 'pq_loss_and_codes']
