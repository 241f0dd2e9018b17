"""MlpUpscaler (reference: VQCPCB/upscalers/mlp_upscaler.py:4-34): Linear -> Dropout -> SELU -> Linear."""
from torch import nn

from .. import ops
from ..utils import SEEDS


class MlpUpscaler(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_size, dropout):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.p = dropout
        # same container / state_dict keys as the reference: mlp.0.{weight,bias}, mlp.3.{weight,bias}
        self.mlp = nn.Sequential(nn.Linear(input_dim, hidden_size, bias=True), nn.Dropout(p=dropout), nn.SELU(),
                                 nn.Linear(hidden_size, output_dim, bias=True))

    def forward(self, inputs):
        p = self.p if self.training else 0.0
        h = ops.linear(inputs, self.mlp[0].weight, self.mlp[0].bias)
        h = ops.DropoutSeluFn.apply(h, p, SEEDS.next() if p > 0 else 0)
        return ops.linear(h, self.mlp[3].weight, self.mlp[3].bias)
