"""Dense feed-forward blocks (reference: ``naive_gpt/layers/basic/feedforward.py``):
``Feedforward`` = fc2(act(dropout(fc1 x))) (OPT/BERT), ``LLaMaFeedforward`` =
down(act(gate x) * side x), no biases."""
import torch
from torch import nn


class Feedforward(nn.Module):
    def __init__(self, d_model: int, d_feedforward: int, p_dropout: float,
                 activation: nn.Module):
        super().__init__()
        self.d_model = d_model
        self.d_feedforward = d_feedforward
        self.p_dropout = p_dropout
        self.fc1 = nn.Linear(d_model, d_feedforward)
        self.fc2 = nn.Linear(d_feedforward, d_model)
        self.dropout = nn.Dropout(p=p_dropout)
        self.activation = activation

    def forward(self, x: torch.Tensor):
        # dropout sits between fc1 and the activation (reference: feedforward.py:30-35)
        return self.fc2(self.activation(self.dropout(self.fc1(x))))


class LLaMaFeedforward(nn.Module):
    def __init__(self, d_model: int, d_feedforward: int, activation: nn.Module):
        super().__init__()
        self.d_model = d_model
        self.d_feedforward = d_feedforward
        self.gate = nn.Linear(d_model, d_feedforward, bias=False)
        self.side = nn.Linear(d_model, d_feedforward, bias=False)
        self.down = nn.Linear(d_feedforward, d_model, bias=False)
        self.activation = activation

    def forward(self, x: torch.Tensor):
        return self.down(self.activation(self.gate(x)) * self.side(x))
