"""Update-operator building blocks (reference: ramp/blocks.py:15-50).

SoftAgg's segment softmax / sum runs on the HIP segment kernel
(csrc/graph.hip) over a device-side group_by instead of torch.unique +
torch_scatter; GatedResidual is three Linear layers (hipBLASLt through torch)."""
import torch
import torch.nn as nn

from . import ops


class GatedResidual(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gate = nn.Sequential(nn.Linear(dim, dim), nn.Sigmoid())
        self.res = nn.Sequential(nn.Linear(dim, dim), nn.ReLU(inplace=True), nn.Linear(dim, dim))

    def forward(self, x):
        return x + self.gate(x) * self.res(x)


class SoftAgg(nn.Module):
    def __init__(self, dim=512, expand=True):
        super().__init__()
        self.dim = dim
        self.expand = expand
        self.f = nn.Linear(dim, dim)
        self.g = nn.Linear(dim, dim)
        self.h = nn.Linear(dim, dim)

    def forward(self, x, ix, groups=None, max_groups=None):
        """x [1,E,dim], ix [E] int64 group keys.  ``groups`` / ``max_groups`` let the
        caller reuse a group_by plan and skip the size read-back (the reference
        synchronises in torch.unique)."""
        if groups is None:
            groups = ops.group_by(ix)
        if max_groups is None:
            max_groups = int(groups.ngroups.item())
        y = ops.segment_softmax_sum(self.f(x)[0], self.g(x)[0], groups, max_groups)
        hy = self.h(y)
        if self.expand:
            return hy[groups.gid[:x.shape[1]].long()].unsqueeze(0)
        return hy.unsqueeze(0)
