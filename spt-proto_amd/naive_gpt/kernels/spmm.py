"""kernels.spmm -- batched CSR x dense.

Mirror of the reference's ``naive_gpt/kernels/spmm.py:6-58``:
``y[b] = A[b] . x[b]``; backward ``grad_values = sddmm(dY, x)`` and
``grad_x = A^T . dY``.  Single stream, see ``kernels/sddmm.py``.
"""
import torch

from naive_gpt import ext


class SPMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, indptr, indices, values, x):
        ctx.save_for_backward(indptr, indices, values, x)
        return ext.spmm_forward_cuda(False, False, indptr, indices, values, x)

    @staticmethod
    def backward(ctx, grad_output: torch.Tensor):
        indptr, indices, values, x = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        grad_values = grad_x = None
        if ctx.needs_input_grad[2]:
            grad_values = ext.sddmm_forward_cuda(
                False, True, indptr, indices, grad_output, x
            )
        if ctx.needs_input_grad[3]:
            grad_x = ext.spmm_transposed(
                ext.transposed_for(indptr, indices, grad_output.size(-1)),
                indptr, indices, values, grad_output
            )
        return None, None, grad_values, grad_x


def spmm(indptr: torch.Tensor, indices: torch.Tensor,
         values: torch.Tensor, x: torch.Tensor):
    return SPMM.apply(indptr, indices, values, x)
