"""kernels.lookup -- PQ-code top-k key selection (integer op, no gradient).

Mirror of the reference's ``naive_gpt/kernels/lookup.py:6-24``.  The sparsity
coefficient travels as the *length* of a CPU tensor (``lookup.cu:99``); that odd
convention is part of the ``ext`` boundary and is kept.
"""
import torch

from naive_gpt import ext


class Lookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, config: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        return ext.lookup_forward_cuda(config, query, key)

    @staticmethod
    def backward(ctx, grad_output: torch.Tensor):
        # integer inputs: never differentiated (reference: lookup.py:15-17)
        raise NotImplementedError


def lookup(query: torch.Tensor, key: torch.Tensor, sparse_coeff: int):
    config = torch.empty([sparse_coeff])
    return Lookup.apply(config, query, key)
