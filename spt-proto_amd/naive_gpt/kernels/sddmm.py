"""kernels.sddmm -- sampled dense-dense matmul over a batched CSR pattern.

Mirror of the reference's ``naive_gpt/kernels/sddmm.py:6-60``:
``out[b,p] = <query[b,row(p)], key[b,indices[b,p]]>``; the backward is two SpMMs,
``grad_query = dA.K`` and ``grad_key = dA^T.Q``.

The reference forks a second CUDA stream for the transposed SpMM
(``sddmm.py:43-50``).  On MI355X each of the two kernels already occupies all 256
CUs (one 1024-thread workgroup per CU), so both are enqueued on the current stream.
The transposed product is a gather over the transposed CSR structure, built once per
pattern and shared by every A^T product of the backward pass (``ext.transposed_for``).
"""
import torch

from naive_gpt import ext


class SDDMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, indptr, indices, query, key):
        ctx.save_for_backward(indptr, indices, query, key)
        return ext.sddmm_forward_cuda(False, True, indptr, indices, query, key)

    @staticmethod
    def backward(ctx, grad_output: torch.Tensor):
        indptr, indices, query, key = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        grad_query = grad_key = None
        if ctx.needs_input_grad[2]:
            grad_query = ext.spmm_forward_cuda(
                False, False, indptr, indices, grad_output, key
            )
        if ctx.needs_input_grad[3]:
            grad_key = ext.spmm_transposed(
                ext.transposed_for(indptr, indices, query.size(-1)),
                indptr, indices, grad_output, query
            )
        return None, None, grad_query, grad_key


def sddmm(indptr: torch.Tensor, indices: torch.Tensor,
          query: torch.Tensor, key: torch.Tensor):
    return SDDMM.apply(indptr, indices, query, key)
