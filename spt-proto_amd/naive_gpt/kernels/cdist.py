"""kernels.cdist -- autograd wrapper of the PQ distance operator.

Mirror of the reference's ``naive_gpt/kernels/cdist.py:6-30``: forward returns
``(distance [M,NQ,C], indices [M,NQ] int32)``; only ``distance`` carries gradient
(``extension/cdist.cu:252-333``).
"""
import torch

from naive_gpt import ext


class CDist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, query: torch.Tensor, table: torch.Tensor):
        distance, indices = ext.cdist_forward_cuda(query, table)
        ctx.save_for_backward(query, table)
        ctx.mark_non_differentiable(indices)
        return distance, indices

    @staticmethod
    def backward(ctx, grad_distance: torch.Tensor, _grad_indices):
        query, table = ctx.saved_tensors
        grad_query, grad_table = ext.cdist_backward_cuda(
            query, table, grad_distance.contiguous()
        )
        return grad_query, grad_table


def cdist(query: torch.Tensor, table: torch.Tensor):
    return CDist.apply(query, table)
