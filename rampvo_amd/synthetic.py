"""Deterministic synthetic event+frame streams and network weights.

There are no datasets or checkpoints offline, so benchmarks and parity tests run
on (a) a seeded synthetic stream shaped like the reference's evaluation input
(``evaluate.py:232-260``: events [1,1,5,H,W], image [1,1,3,H,W], mask [1],
intrinsics [4]) and (b) seeded random-init weights of the reference architecture.

Stream: a smooth random texture on a fronto-parallel plane seen by a camera that
translates sideways (so image flow is a few px/frame at feature resolution and
the tracker's motion probe / keyframe logic behave as on real data).  Events are
a 5-bin signed temporal difference of the texture with continuous values (so the
top-k patch selection is tie-free).

Weights: default torch initialisers under a fixed seed, with the ``d`` head of the
update operator scaled so that random weights produce >= 2 px median updates --
with plain random init the motion probe of ``Ramp_vo`` never fires and the tracker
would not initialise ("No checkpoints offline").  Two profiles:

  * ``wide`` (default; the throughput workload): large, noisy updates -- keyframes are
    kept, the sliding window fills to its bound (E ~ 40k of 45k edges at configs[1]).
    A random-weight tracker in this regime is a chaotic feedback loop (BA fits 2-40 px
    of inconsistent residuals with full confidence; depths hit the ``d > 20 -> 1``
    reset), so float parity is checked teacher-forced, one ``update()`` at a time.
  * ``damped`` (trajectory parity): unit ``d`` gain and confidence logits shifted by
    -16 (weights ~1e-7): Gauss-Newton is dominated by its damping terms, one
    ``update()`` no longer amplifies a perturbation, and an fp32 free run stays
    within 1e-4 of the reference run over tens of frames (tests/golden/ramp_vo_traj_*).
"""
import math

import torch
import torch.nn.functional as F


def _smooth_noise(h, w, gen, octaves=((4, 1.0), (12, 0.6), (40, 0.35), (120, 0.2))):
    tex = torch.zeros(1, 1, h, w)
    for cells, amp in octaves:
        gh, gw = max(2, h * cells // 480), max(2, w * cells // 480)
        g = torch.randn(1, 1, gh, gw, generator=gen)
        tex = tex + amp * F.interpolate(g, size=(h, w), mode="bicubic", align_corners=False)
    return tex[0, 0]


class SyntheticStream:
    """iterable of (image, events, intrinsics, mask) like evaluate._data_iterator"""

    def __init__(self, H=480, W=640, T=200, seed=1234, speed=(3.1, 1.3), mask_every=1, device="cpu"):
        self.H, self.W, self.T, self.device = H, W, T, torch.device(device)
        gen = torch.Generator().manual_seed(seed)
        self.margin = int(math.ceil(max(abs(speed[0]), abs(speed[1])) * (T + 2))) + 8
        ch, cw = H + 2 * self.margin, W + 2 * self.margin
        self.canvas = torch.stack([_smooth_noise(ch, cw, gen) for _ in range(3)], 0).to(self.device)   # [3,ch,cw]
        self.canvas = self.canvas / self.canvas.abs().max()
        self.speed = speed
        self.mask_every = mask_every
        fx = 320.0 * W / 640.0
        self.intrinsics = torch.tensor([fx, fx, W / 2.0, H / 2.0])

    def _view(self, t):
        """sub-pixel crop of the canvas at (fractional) time t -> [3,H,W]"""
        x0 = self.margin + self.speed[0] * t
        y0 = self.margin + self.speed[1] * t
        ix, iy = int(math.floor(x0)), int(math.floor(y0))
        fx, fy = x0 - ix, y0 - iy
        c = self.canvas
        H, W = self.H, self.W
        a = c[:, iy:iy + H, ix:ix + W]
        b = c[:, iy:iy + H, ix + 1:ix + 1 + W]
        cc = c[:, iy + 1:iy + 1 + H, ix:ix + W]
        d = c[:, iy + 1:iy + 1 + H, ix + 1:ix + 1 + W]
        return (1 - fy) * ((1 - fx) * a + fx * b) + fy * ((1 - fx) * cc + fx * d)

    def frame(self, t):
        img = self._view(float(t))
        image = (img * 0.9 + 0.25).reshape(1, 1, 3, self.H, self.W)          # roughly the [-0.5, 1.5) range
        bins = []
        for b in range(5):
            d = self._view(t + (b + 1) / 5.0 - 1.0) - self._view(t + b / 5.0 - 1.0)
            bins.append(d.mean(0))
        events = (torch.stack(bins, 0) * 40.0).reshape(1, 1, 5, self.H, self.W)
        mask = torch.tensor([(t % self.mask_every) == 0])
        return image.float(), events.float(), self.intrinsics.clone(), mask

    def __len__(self):
        return self.T

    def __iter__(self):
        for t in range(self.T):
            yield self.frame(t)


WEIGHT_PROFILES = {
    "wide": dict(d_gain=20.0, d_bias=2.0, w_bias=0.0),
    "damped": dict(d_gain=1.0, d_bias=2.0, w_bias=-16.0),
}


def seeded_state_dict(net_module, seed=1234, d_gain=None, d_bias=None, w_bias=None, profile="wide"):
    """deterministic weights for a (reference-named) VONet: torch default inits under
    ``seed`` + a scaled ``update.d`` head (+ a shifted ``update.w`` bias).  Returns a CPU
    state_dict; the same dict is loaded into the reference network when golden vectors
    are generated."""
    prof = WEIGHT_PROFILES[profile]
    d_gain = prof["d_gain"] if d_gain is None else d_gain
    d_bias = prof["d_bias"] if d_bias is None else d_bias
    w_bias = prof["w_bias"] if w_bias is None else w_bias
    torch.manual_seed(seed)
    for m in net_module.modules():
        if hasattr(m, "reset_parameters"):
            m.reset_parameters()
    for m in net_module.modules():
        if isinstance(m, torch.nn.Conv2d) and m.kernel_size != (1, 1):
            torch.nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
    sd = {k: v.detach().clone().cpu() for k, v in net_module.state_dict().items()}
    sd["update.d.1.weight"] = sd["update.d.1.weight"] * d_gain
    sd["update.d.1.bias"] = sd["update.d.1.bias"] + d_bias
    sd["update.w.1.bias"] = sd["update.w.1.bias"] + w_bias
    return sd


def make_network(input_mode="SingleScale", seed=1234, device="cuda", **kw):
    from .net import VONet
    net = VONet({"event_bias": True, "num_event_bins": 5, "input_mode": input_mode})
    net.load_state_dict(seeded_state_dict(net, seed, **kw))
    return net.to(device).eval()
