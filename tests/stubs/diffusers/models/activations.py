"""diffusers.models.activations (0.30.2): get_activation, GEGLU, GELU, ApproximateGELU."""
import torch.nn.functional as F
from torch import nn

_ACT = {"swish": nn.SiLU, "silu": nn.SiLU, "mish": nn.Mish, "gelu": nn.GELU, "relu": nn.ReLU}


def get_activation(act_fn):
    return _ACT[act_fn.lower()]()


class GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, hidden_states, *args, **kwargs):
        return F.gelu(self.proj(hidden_states), approximate=self.approximate)


class GEGLU(nn.Module):
    """proj to 2*dim_out, first half = value, second half = gate: value * gelu_erf(gate)"""

    def __init__(self, dim_in, dim_out, bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2, bias=bias)

    def forward(self, hidden_states, *args, **kwargs):
        hidden_states = self.proj(hidden_states)
        hidden_states, gate = hidden_states.chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)


class ApproximateGELU(nn.Module):
    def __init__(self, dim_in, dim_out, bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)

    def forward(self, x, *args, **kwargs):
        x = self.proj(x)
        return x * (1.702 * x).sigmoid()
