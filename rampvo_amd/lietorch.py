"""SE3 subset of the reference's lietorch python layer (ramp/lietorch/groups.py:51-312),
forward only, float32, backed by the HIP kernels of csrc/lie.hip."""
import numpy as np
import torch

from . import ops


class SE3:
    group_name = 'SE3'
    group_id = 3
    manifold_dim = 6
    embedded_dim = 7
    id_elem = torch.as_tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])

    def __init__(self, data):
        self.data = data

    def __repr__(self):
        return "SE3: size={}, device={}, dtype={}".format(self.shape, self.device, self.dtype)

    @property
    def shape(self):
        return self.data.shape[:-1]

    @property
    def device(self):
        return self.data.device

    @property
    def dtype(self):
        return self.data.dtype

    @classmethod
    def Identity(cls, *batch_shape, **kwargs):
        if isinstance(batch_shape[0], (tuple, list)):
            batch_shape = tuple(batch_shape[0])
        data = cls.id_elem.to(device=kwargs.get("device", "cpu"), dtype=kwargs.get("dtype", torch.float32))
        return cls(data.repeat(int(np.prod(batch_shape)), 1).view(tuple(batch_shape) + (7,)))

    @classmethod
    def IdentityLike(cls, G):
        return cls.Identity(G.shape, device=G.data.device, dtype=G.data.dtype)

    @classmethod
    def exp(cls, x):
        return cls(ops.se3_unary("ramp_se3_exp", x, 6, 7))

    def log(self):
        return ops.se3_unary("ramp_se3_log", self.data, 7, 6)

    def inv(self):
        return SE3(ops.se3_unary("ramp_se3_inv", self.data, 7, 7))

    def mul(self, other):
        return SE3(ops.se3_binary("ramp_se3_mul", self.data, other.data, 7, 7, 7))

    def retr(self, a):
        return SE3.exp(a).mul(self)

    def adj(self, a):
        return ops.se3_binary("ramp_se3_adj", self.data, a, 7, 6, 6)

    def adjT(self, a):
        return ops.se3_binary("ramp_se3_adjT", self.data, a, 7, 6, 6)

    def act(self, p):
        if p.shape[-1] == 3:
            p4 = torch.cat([p, torch.ones_like(p[..., :1])], dim=-1)
            return ops.se3_binary("ramp_se3_act4", self.data, p4, 7, 4, 4)[..., :3]
        return ops.se3_binary("ramp_se3_act4", self.data, p, 7, 4, 4)

    def matrix(self):
        I = torch.eye(4, dtype=self.dtype, device=self.device)
        I = I.view([1] * (len(self.data.shape) - 1) + [4, 4])
        return SE3(self.data[..., None, :]).act(I).transpose(-1, -2)

    def translation(self):
        p = torch.as_tensor([0.0, 0.0, 0.0, 1.0], dtype=self.dtype, device=self.device)
        p = p.view([1] * (len(self.data.shape) - 1) + [4, ])
        return self.act(p)

    def scale(self, s):
        t, q = self.data.split([3, 4], -1)
        return SE3(torch.cat([t * s.unsqueeze(-1), q], dim=-1))

    def detach(self):
        return SE3(self.data.detach())

    def view(self, dims):
        return SE3(self.data.view(tuple(dims) + (7,)))

    def __mul__(self, other):
        if isinstance(other, SE3):
            return self.mul(other)
        if isinstance(other, torch.Tensor):
            return self.act(other)
        return NotImplemented

    def __getitem__(self, index):
        return SE3(self.data[index])

    def __setitem__(self, index, item):
        self.data[index] = item.data

    def to(self, *args, **kwargs):
        return SE3(self.data.to(*args, **kwargs))

    def cpu(self):
        return SE3(self.data.cpu())

    def cuda(self):
        return SE3(self.data.cuda())


def cat(group_list, dim):
    return SE3(torch.cat([X.data for X in group_list], dim=dim))


def stack(group_list, dim):
    return SE3(torch.stack([X.data for X in group_list], dim=dim))
