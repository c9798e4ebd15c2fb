"""Small helper modules (reference: ``naive_gpt/layers/basic/utils.py``)."""
import torch
from torch import nn


class FnModule(nn.Module):
    """Wrap a plain function (and bound extra arguments) as a module
    (reference: ``utils.py:5-19``)."""

    def __init__(self, fn: callable, *args, **kwargs):
        super().__init__()
        self.fn = fn
        self.args = args
        self.kwargs = kwargs

    def forward(self, *args, **kwargs):
        return self.fn(*args, *self.args, **kwargs, **self.kwargs)


class LlamaRMSNorm(nn.Module):
    """RMS normalisation with a learned gain (reference: ``utils.py:22-37``);
    statistics in fp32, output in the gain's dtype when that is 16-bit."""

    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x: torch.Tensor):
        mean_sq = x.float().square().mean(dim=-1, keepdim=True)
        x = x * torch.rsqrt(mean_sq + self.variance_epsilon)
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            x = x.to(self.weight.dtype)
        return self.weight * x


# a pure function of its input: may be recomputed in the backward (layers/tuning/recompute.py)
from ..tuning import recompute as _recompute  # noqa: E402

_recompute.register(LlamaRMSNorm)
