"""Product quantiser (reference: ``naive_gpt/layers/basic/quantizer.py:6-139``).

``forward(mode, z)`` with ``mode`` in

* ``'encode'``   -> PQ codes ``[..., M]`` (int64 for v1 / int32 for v2, as the reference),
* ``'decode'``   -> centroids looked up from codes,
* ``'quantize'`` -> nearest centroids ``z_q`` with the shape of ``z``,
* ``'train'``    -> ``(z_q, loss)``,
  ``loss = MSE(softmax(-log clamp(d, 1e-5)) . W, z_q) + MSE(z, z_q)``.

The codebook ``weight [M, C, D]`` is ``randn``-initialised.  ``method='v1'`` measures
with ``torch.cdist(p=1)`` + ``argmin``; ``method='v2'`` calls the HIP operator
``kernels.cdist`` (L1 distance + fused argmin, ``spt-proto_amd/csrc/cdist.hip``).
In ``'encode'`` mode v2 uses the indices-only entry of the same kernel, which skips
the ``[M, NQ, C]`` distance tensor the reference writes and throws away.
"""
import torch
from torch import nn

from naive_gpt import ext, kernels

_MODES = ('train', 'encode', 'decode', 'quantize')


class PQBase(nn.Module):
    def __init__(self, d_codeword: int, n_codewords: int, n_subspaces: int, method: str):
        super().__init__()
        self.method = method
        self.d_codeword = d_codeword
        self.n_codewords = n_codewords
        self.n_subspaces = n_subspaces
        self.weight = nn.Parameter(torch.randn([n_subspaces, n_codewords, d_codeword]))
        self.loss_fn = nn.MSELoss()

    def _subspace_major(self, z: torch.Tensor):
        # [..., M*D] -> [M, NQ, D] (the layout of the cdist operator)
        flat = z.flatten(end_dim=-2)
        flat = flat.view([flat.size(0), self.n_subspaces, -1])
        return flat.transpose(0, 1).contiguous()

    def _nearest(self, z_flat: torch.Tensor, want_distance: bool):
        """-> (distance [M,NQ,C] | None, indices [M,NQ,1])"""
        if self.method == 'v1':
            distance = torch.cdist(
                z_flat.float(), self.weight.float(), p=1.0
            ).type(z_flat.dtype)
            return distance, torch.argmin(distance, dim=-1, keepdim=True)
        if self.method == 'v2':
            if not want_distance:
                with torch.no_grad():
                    indices = ext.cdist_encode(z_flat.detach(), self.weight.detach())
                return None, indices.unsqueeze(-1)
            distance, indices = kernels.cdist(z_flat, table=self.weight)
            return distance, indices.unsqueeze(-1)
        raise RuntimeError

    def train_loss(self, z: torch.Tensor) -> torch.Tensor:
        """``self('train', z)[-1]``.  For the v2 quantiser on the GPU this is the fused
        operator ``kernels.pq_loss`` (two kernels, no [M, NQ, C] intermediates); any
        other case composes the loss exactly as the reference does."""
        if self.method == 'v2' and ext.pq_loss_supported(z, self.weight):
            assert z.size(-1) == self.d_codeword * self.n_subspaces
            return kernels.pq_loss(z, self.weight)
        return self('train', z=z)[-1]

    def train_loss_and_codes(self, z: torch.Tensor):
        """(``self('train', z)[-1]``, PQ codes [N * H, S, M]) of z [N, S, H, E] in one pass where
        the fused operator applies (its argmin is the code), else (loss, None)."""
        if self.method == 'v2' and z.dim() == 4 and ext.pq_loss_supported(z, self.weight):
            return kernels.pq_loss_and_codes(z, self.weight)
        return self.train_loss(z), None

    def forward(self, mode: str, z: torch.Tensor):
        assert mode in _MODES
        assert z.dim() > 1
        if mode == 'decode':
            assert z.size(-1) == self.n_subspaces
        else:
            assert z.size(-1) == self.d_codeword * self.n_subspaces
        out_shape = list(z.size())[:-1] + [-1]
        z_flat = self._subspace_major(z)

        distance = None
        if mode == 'decode':
            indices = z_flat
        else:
            distance, indices = self._nearest(z_flat, want_distance=(mode == 'train'))
        assert indices.dim() == 3

        if mode == 'encode':
            return indices.transpose(0, 1).view(out_shape).contiguous()

        # hard centroids
        gather_index = indices.long().expand(size=[-1, -1, self.d_codeword])
        z_q_flat = torch.gather(self.weight, dim=1, index=gather_index)
        z_q = torch.reshape(z_q_flat.transpose(0, 1), shape=out_shape)
        if mode in ('decode', 'quantize'):
            return z_q
        if mode != 'train':
            raise RuntimeError

        # soft assignment by inverse distance pulls the codebook, the second term
        # pulls the inputs to their centroids (reference: quantizer.py:97-111)
        distance = torch.clamp(distance, min=1e-5)
        soft = torch.softmax(-torch.log(distance), dim=-1)
        z_w = torch.matmul(soft, self.weight)
        loss = self.loss_fn(z_w, target=z_q_flat) + self.loss_fn(z_flat, target=z_q_flat)
        return z_q, loss


class PQV1(PQBase):
    def __init__(self, d_codeword: int, n_codewords: int, n_subspaces: int):
        super().__init__(d_codeword=d_codeword, n_codewords=n_codewords,
                         n_subspaces=n_subspaces, method='v1')


class PQV2(PQBase):
    def __init__(self, d_codeword: int, n_codewords: int, n_subspaces: int):
        super().__init__(d_codeword=d_codeword, n_codewords=n_codewords,
                         n_subspaces=n_subspaces, method='v2')
