"""kernels.softmax -- causal-masked softmax over CSR rows.

Mirror of the reference's ``naive_gpt/kernels/softmax.py:6-38``; the backward
consumes the saved *output* (``extension/softmax.cu:49-81``).
"""
import torch

from naive_gpt import ext


class Softmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, indptr, indices, values):
        output = ext.softmax_forward_cuda(indptr, indices, values)
        ctx.save_for_backward(indptr, indices, output)
        return output

    @staticmethod
    def backward(ctx, grad_output: torch.Tensor):
        indptr, indices, output = ctx.saved_tensors
        grad_values = ext.softmax_backward_cuda(
            indptr, indices, output, grad_output.contiguous()
        )
        return None, None, grad_values


def softmax(indptr: torch.Tensor, indices: torch.Tensor, values: torch.Tensor):
    return Softmax.apply(indptr, indices, values)
