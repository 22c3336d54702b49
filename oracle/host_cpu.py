"""Torch (ATen) forwards of the network modules and the CPU variant of the tracker -- TEST / CPU-BASELINE ONLY.

``rampvo_amd`` runs its encoder, update operator and tracker state machine on HIP kernels or raises.  The host
logic is exercised without a GPU -- and the "port" CPU baseline of ``bench.py`` is timed -- through this module:

  * plain-PyTorch restatements of the module forwards (reference ramp/extractor.py, ramp/net.py:69-90) that
    ``oracle.backend_cpu.cpu_oracle_ops()`` binds to the product's module classes for the duration of the context;
    they also serve, on GPU tensors, as the fp32 torch reference of the encoder kernels in tests/test_encoder_gpu.py;
  * ``RampVoCPU``: ``rampvo_amd.Ramp_vo.Ramp_vo`` with the device-side steps replaced by oracle-backed ones; the
    graph bookkeeping, edge generation, keyframe logic and pose-prediction code are the product's own.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from rampvo_amd import altcorr, fastba, ops
from rampvo_amd import projective_ops as pops
from rampvo_amd._lib import RAMP_NHWC
from rampvo_amd.lietorch import SE3
from rampvo_amd.net import GraphPlan
from rampvo_amd.Ramp_vo import Ramp_vo
from rampvo_amd.utils import Timer, filter_features


# ------------------------------------------------------------------ conv towers
def conv_norm_relu(x, conv, norm, relu, out_scale=1.0):
    """y = [relu]([instance_norm](conv(x) + b)) * out_scale"""
    x = x.contiguous(memory_format=torch.channels_last)
    y = F.conv2d(x.to(conv.weight.dtype), conv.weight, conv.bias, conv.stride, conv.padding)
    if isinstance(norm, nn.InstanceNorm2d):
        y = F.instance_norm(y, eps=norm.eps)
    elif not (norm is None or (isinstance(norm, nn.Sequential) and len(norm) == 0)):
        y = norm(y)
    if relu:
        y = F.relu(y, inplace=True)
    if out_scale != 1.0:
        y = y * out_scale
    return y


def residual_block_forward(self, x):                                   # reference extractor.py:49-57
    y = conv_norm_relu(x, self.conv1, self.norm1, relu=True)
    y = conv_norm_relu(y, self.conv2, self.norm2, relu=True)
    if self.downsample is not None:
        x = conv_norm_relu(x, self.downsample[0], self.norm3, relu=False)
    return F.relu(x + y, inplace=True)


def basic_encoder4_forward(self, x, out_scale=1.0):                     # reference extractor.py:112-126
    b, n, c1, h1, w1 = x.shape
    x = x.reshape(b * n, c1, h1, w1)
    x = conv_norm_relu(x, self.conv1, self.norm1, relu=True)
    x = self.layer1(x)
    x = self.layer2(x)
    x = conv_norm_relu(x, self.conv2, None, relu=False, out_scale=out_scale)
    return x.view(b, n, *x.shape[1:])


def _cat(x, y):
    return torch.cat((x, y.to(x.dtype)), dim=1).contiguous(memory_format=torch.channels_last)


def multiscale_encoder4_forward(self, x, x_down2, x_down4, out_scale=1.0):   # reference extractor.py:288-311
    b, n = x.shape[:2]
    x = x.reshape(b * n, *x.shape[2:])
    x_down2 = x_down2.reshape(b * n, *x_down2.shape[2:])
    x_down4 = x_down4.reshape(b * n, *x_down4.shape[2:])
    x = conv_norm_relu(x, self.conv1, self.norm1, relu=True)
    x = self.layer1(x)
    x = _cat(x, x_down2)
    x = self.layer3(x)
    x = _cat(x, x_down4)
    x = conv_norm_relu(x, self.conv3, None, relu=False, out_scale=out_scale)
    return x.view(b, n, *x.shape[1:])


# ------------------------------------------------------------------ per-pixel recurrent cells
def _pixel_lstm(lstm, x2d, state):
    """one step of nn.LSTM (gate order i,f,g,o) for every pixel row of x2d [HW,Cin]; state = (h, c) or None (zeros)"""
    gates = torch.addmm(lstm.bias_ih_l0 + lstm.bias_hh_l0, x2d, lstm.weight_ih_l0.t())
    if state is not None:
        gates = gates + state[0] @ lstm.weight_hh_l0.t()
    i, f, g, o = gates.chunk(4, dim=1)
    i, f, o = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o)
    g = torch.tanh(g)
    c = i * g if state is None else f * state[1] + i * g
    h = o * torch.tanh(c)
    return h, c


def _pixel_mix(conv, s2d, e2d):
    """1x1 conv on the channel concat [s ; e] as two [HW,C] GEMMs"""
    w = conv.weight.view(conv.out_channels, -1)
    k = s2d.shape[1]
    return torch.addmm(conv.bias, s2d, w[:, :k].t()) + e2d @ w[:, k:].t()


def _to_rows(x):
    return x.permute(1, 2, 0).reshape(-1, x.shape[0]).contiguous()


def merger_forward(self, events, images, reinit_hidden=False, out_scale=1.0):    # reference extractor.py:233-269
    B, T, Ce, H, W = events.shape
    assert B == 1 and images.shape[1] == T
    if reinit_hidden:
        self.states_events, self.states_image, self.super_state = None, None, None
    super_states = []
    for t in range(T):
        ev, im = events[0, t], images[0, t]
        e_rows, i_rows = _to_rows(ev.float()), _to_rows(im.float())
        self.states_events = _pixel_lstm(self.events_convlstm, e_rows, self.states_events)
        self.states_image = _pixel_lstm(self.image_convlstm, i_rows, self.states_image)
        s = self.super_state if self.super_state is not None else torch.zeros_like(self.states_events[0])
        s_ev = _pixel_mix(self.superstate_encoder, s, self.states_events[0])
        s = torch.where((ev != 0).any(), s_ev, s)
        s_im = _pixel_mix(self.superstate_encoder, s, self.states_image[0])
        s = torch.where((im != 0).any(), s_im, s)
        self.super_state = s
        super_states.append(s.view(H, W, -1))
    ss = torch.stack(super_states, 0).permute(0, 3, 1, 2)[None]
    fmap = basic_encoder4_forward(self.fmap_encoder, ss, out_scale=out_scale)
    imap = basic_encoder4_forward(self.imap_encoder, ss, out_scale=out_scale)
    return fmap, imap, None


def lstm_encoder_forward(self, x):
    """reference extractor.py:376-385: conv_1, then nn.LSTM over the T steps of the call with h0 = c0 = 0 (the
    state is threaded through the T steps of ONE call, not across calls) -> list over T of [H'*W', hid]"""
    y = conv_norm_relu(x[0].float(), self.conv_1, None, relu=False)
    Hs, Ws = y.shape[-2:]
    out, state = [], None
    for t in range(y.shape[0]):
        state = _pixel_lstm(self.convlstm, _to_rows(y[t]), state)
        out.append(state[0])
    return out, (Hs, Ws)


def multiscale_forward(self, events, images, mask, reinit_hidden=False, out_scale=1.0):   # reference extractor.py:540-566
    mask_list = [bool(m) for m in mask.reshape(-1).tolist()]
    outs = []
    for k in range(len(self.scales)):
        if reinit_hidden:
            self.super_states[k] = None
        ev_rows, (Hs, Ws) = lstm_encoder_forward(self.ev_encoders[k], events)
        im_rows, _ = lstm_encoder_forward(self.im_encoders[k], images)
        s = self.super_states[k]
        collected, ind_im = [], 0
        for t, e in enumerate(ev_rows):
            if s is None:
                s = torch.zeros_like(e)
            s = _pixel_mix(self.super_state_ev_encoder[k].encoder, s, e)
            if mask_list[t] if len(mask_list) > 1 else mask_list[0]:
                s = _pixel_mix(self.super_state_im_encoders[k].encoder, s, im_rows[ind_im])
                ind_im += 1
                collected.append(s)
        self.super_states[k] = s
        stack = collected if collected else [s]
        outs.append(torch.stack([r.view(Hs, Ws, -1) for r in stack], 0).permute(0, 3, 1, 2)[None])
    fmap = multiscale_encoder4_forward(self.fmap_encoder, outs[0], outs[1], outs[2], out_scale=out_scale)
    imap = multiscale_encoder4_forward(self.imap_encoder, outs[0], outs[1], outs[2], out_scale=out_scale)
    return fmap, imap


# ------------------------------------------------------------------ update operator
def update_forward(self, net, inp, corr, flow, ii, jj, kk, plan=None):       # reference net.py:69-90
    if plan is None:
        plan = GraphPlan.build(ii, jj, kk)
    mask_ix = (plan.ix_raw >= 0).reshape(1, -1, 1)
    mask_jx = (plan.jx_raw >= 0).reshape(1, -1, 1)
    ix, jx = plan.ix_raw.clamp(min=0), plan.jx_raw.clamp(min=0)
    net = net + inp + self.corr(corr)
    net = self.norm(net)
    net = net + self.c1(mask_ix.to(net.dtype) * net[:, ix])
    net = net + self.c2(mask_jx.to(net.dtype) * net[:, jx])
    net = net + self.agg_kk(net, kk, plan.g_kk, plan.max_kk)
    net = net + self.agg_ij(net, None, plan.g_ij, plan.max_ij)
    net = self.gru(net)
    return net, (self.d(net), self.w(net), None)


# ------------------------------------------------------------------ update operator, fp16 policy
# The SAME expressions with the roundings the shipped MIXED_PRECISION path applies (csrc/update_mlp.hip,
# csrc/update.hip; the reference's autocast makes the same tensors half, ramp/net.py:69-90 under
# torch.amp.autocast): Linear operands (input, weight, bias) and outputs are fp16 values, products accumulate in
# fp32; LayerNorm, the residual stream, the segment softmax and every sum stay fp32; sigmoid(gate) is a half tensor.
# With this leg a full-size step separates KERNEL error (HIP fp16 vs this) from PRECISION-POLICY error (this vs the
# fp32 oracle): tests/test_pipeline_gpu.py::_one_update_vs_cpu_oracle(policy=True).
def _h(x):
    return x.half().float()


def _lin16(lin, x):
    return _h(F.linear(_h(x), _h(lin.weight), _h(lin.bias)))


def _gated16(gr, x):
    gate = _h(torch.sigmoid(_lin16(gr.gate[0], x)))
    r = _lin16(gr.res[2], torch.relu(_lin16(gr.res[0], x)))
    return x + gate * r


def _softagg16(agg, x, groups, max_groups):
    f, g = _lin16(agg.f, x[0]), _lin16(agg.g, x[0])
    y = _h(ops.segment_softmax_sum(f, g, groups, max_groups))
    hy = _lin16(agg.h, y)
    return hy[groups.gid[:x.shape[1]].long()].unsqueeze(0)


def update_forward_fp16_policy(self, net, inp, corr, flow, ii, jj, kk, plan=None):
    if plan is None:
        plan = GraphPlan.build(ii, jj, kk)
    mask_ix = (plan.ix_raw >= 0).reshape(1, -1, 1)
    mask_jx = (plan.jx_raw >= 0).reshape(1, -1, 1)
    ix, jx = plan.ix_raw.clamp(min=0), plan.jx_raw.clamp(min=0)
    c = self.corr
    t = torch.relu(_lin16(c[0], _h(corr)))                          # the volume is stored as fp16 (csrc/altcorr.hip)
    t = torch.relu(c[3](_lin16(c[2], t)))
    net = self.norm(net.float() + _h(inp) + _lin16(c[5], t))
    net = net + _lin16(self.c1[2], torch.relu(_lin16(self.c1[0], mask_ix.to(net.dtype) * net[:, ix])))
    net = net + _lin16(self.c2[2], torch.relu(_lin16(self.c2[0], mask_jx.to(net.dtype) * net[:, jx])))
    net = net + _softagg16(self.agg_kk, net, plan.g_kk, plan.max_kk)
    net = net + _softagg16(self.agg_ij, net, plan.g_ij, plan.max_ij)
    net = self.gru[0](net)
    net = _gated16(self.gru[1], net)
    net = self.gru[2](net)
    net = _gated16(self.gru[3], net)
    r = torch.relu(net)
    return net, (_lin16(self.d[1], r), _h(torch.sigmoid(_lin16(self.w[1], r))), None)


def module_patches(fp16_policy=False):
    """(class, attribute, function) triples bound by cpu_oracle_ops()"""
    if fp16_policy:
        return [t if t[2] is not update_forward else (t[0], t[1], update_forward_fp16_policy) for t in module_patches()]
    from rampvo_amd import extractor as ex
    from rampvo_amd import net as netmod
    return [(ex.ResidualBlock, "forward", residual_block_forward),
            (ex.BasicEncoder4, "forward", basic_encoder4_forward),
            (ex.MultiScaleBasicEncoder4, "forward", multiscale_encoder4_forward),
            (ex.MergerLSTMsceneEncoder, "forward", merger_forward),
            (ex.LSTMEncoder, "forward", lstm_encoder_forward),
            (ex.MultiScaleMergerDoubleNet, "forward", multiscale_forward),
            (netmod.Update, "forward", update_forward)]


# ------------------------------------------------------------------ tracker
class RampVoCPU(Ramp_vo):
    """the product's tracker with its device steps served by the CPU oracle (inside cpu_oracle_ops())"""

    _allow_cpu = True

    def __init__(self, cfg, network, train_cfg, ht=480, wd=640, device="cpu"):
        super().__init__(cfg, network, train_cfg, ht=ht, wd=wd, device=device)

    def _layout_flags(self, h, w):
        return False, False            # plain NHWC ring slots; the hidden state is compacted eagerly like the reference's

    def _init_streams(self, dev):
        self._fe_stream = self._ev_fe_done = self._ev_gate = self._ev_in = None
        self._gate_armed = False

    def _current_stream(self):
        return None

    def corr(self, coords, indicies=None, order=None):
        ii, jj = indicies if indicies is not None else (self.kk, self.jj)
        ii1 = ii % (self.M * self.mem)                      # reference Ramp_vo.py:178-179
        jj1 = jj % self.mem
        return altcorr.corr_pyramid(self.gmap_.view(-1, 3, 3, 128), self.pyramid, coords[0], ii1, jj1, 3, (1, 4),
                                    RAMP_NHWC, order=order)

    def reproject(self, indicies=None, poses=None, patches=None, intrinsics=None):
        (ii, jj, kk) = indicies if indicies is not None else (self.ii, self.jj, self.kk)
        poses = poses if poses is not None else self.poses
        patches = patches if patches is not None else self.patches
        intrinsics = intrinsics if intrinsics is not None else self.intrinsics
        return pops.reproject(poses, patches, intrinsics, ii, jj, kk)

    def motion_probe(self):
        kk = torch.arange(self.m - self.M, self.m, device=self.device)
        jj = self.n * torch.ones_like(kk)
        ii = kk // self.M
        coords = self.reproject(indicies=(ii, jj, kk))
        corr = self.corr(coords, indicies=(kk, jj)).to(self.dtype)
        net = torch.zeros(1, len(ii), self.DIM, dtype=torch.float, device=self.device)
        ctx = self.imap[:, kk % (self.M * self.mem)]
        net, (delta, weight, _) = self.network.update(net, ctx, corr, None, ii, jj, kk)
        return torch.quantile(delta.norm(dim=-1).float(), 0.5)

    def _motionmag_pair(self, i, j):
        mags = []
        for a, b in ((i, j), (j, i)):
            sel = np.nonzero((self._ii == a) & (self._jj == b))[0]
            if len(sel) == 0:
                mags.append(torch.full((), float("nan"), device=self.device))
                continue
            s = self._upload(sel)
            flow = pops.flow_mag(self.poses, self.patches, self.intrinsics, self.ii[s], self.jj[s], self.kk[s], beta=0.5)
            mags.append(flow.mean())
        return float(((mags[0] + mags[1]) / 2).item())

    def _graph_edit(self, remove_kf):
        """host-side result of keyframe() for one outcome of the motion test (reference :247-274)"""
        ii, jj, kk = self._ii, self._jj, self._kk
        keep = np.ones(len(ii), bool)
        n = self.n
        if remove_kf:
            k = self.n - self.cfg.KEYFRAME_INDEX
            keep &= ~((ii == k) | (jj == k))
            ii, jj, kk = ii.copy(), jj.copy(), kk.copy()
            kk[ii > k] -= self.M
            ii[ii > k] -= 1
            jj[jj > k] -= 1
            n -= 1
        keep &= ~((kk // self.M) < n - self.cfg.REMOVAL_WINDOW)
        changed = remove_kf or not keep.all()
        idx = np.nonzero(keep)[0] if changed else None
        if changed:
            ii, jj, kk = ii[idx], jj[idx], kk[idx]
        return dict(changed=changed, ii=ii, jj=jj, kk=kk, idx=idx, n=n)

    def _apply_removal(self, k):
        n = self.n
        del self._tstamps[k]
        if self._last_K_row > k:
            self._last_K_row -= 1
        for buf in (self.tstamps_, self.colors_, self.poses_, self.patches_, self.intrinsics_):
            buf[k:n - 1] = buf[k + 1:n].clone()
        dst = torch.arange(k, n - 1, device=self.device) % self.mem
        src = torch.arange(k + 1, n, device=self.device) % self.mem
        for buf in (self.imap_, self.gmap_, self.fmap1_, self.fmap2_):
            buf[dst] = buf[src]
        self.n -= 1
        self.m -= self.M

    def keyframe(self):
        """reference :237-274: both removals decided on the host mirror, applied as ONE compaction"""
        i = self.n - self.cfg.KEYFRAME_INDEX - 1
        j = self.n - self.cfg.KEYFRAME_INDEX + 1
        m = self._motionmag_pair(i, j)
        remove = m < self.cfg.KEYFRAME_THRESH
        if remove:
            k = self.n - self.cfg.KEYFRAME_INDEX
            t0, t1 = self._tstamps[k - 1], self._tstamps[k]
            dP = SE3(self.poses_[k]) * SE3(self.poses_[k - 1]).inv()
            self.delta[t1] = (t0, dP)
        ed = self._graph_edit(remove)
        if remove:
            self._apply_removal(self.n - self.cfg.KEYFRAME_INDEX)
        if not ed["changed"]:
            return
        idx = ed["idx"]
        self._ii, self._jj, self._kk = ed["ii"], ed["jj"], ed["kk"]
        self.ii, self.jj, self.kk = self._upload(self._ii), self._upload(self._jj), self._upload(self._kk)
        if len(idx) != self.net.shape[1]:
            self.net = self.net[:, self._upload(idx)]
        self._plan = None

    def update(self):
        with Timer("other", enabled=self.enable_timing):
            plan = self._graph_plan()
            coords = self.reproject()
            corr = self.corr(coords).to(self.dtype)
            ctx = self.imap[:, self.kk % (self.M * self.mem)]
            self.net, (delta, weight, _) = self.network.update(self.net, ctx, corr, None, self.ii, self.jj, self.kk,
                                                               plan=plan)
            weight = weight.float()
            target = coords[..., self.P // 2, self.P // 2] + delta.float()
            weight = filter_features(confidences=weight, target=target, data_shape=(self.ht // 4, self.wd // 4))
            self.last_weight = weight
        with Timer("BA", enabled=self.enable_timing):
            t0 = self.n - self.cfg.OPTIMIZATION_WINDOW if self.is_initialized else 1
            t0 = max(t0, 1)
            try:
                fastba.BA(self.poses, self.patches, self.intrinsics, target, weight, self.lmbda, self.ii, self.jj,
                          self.kk, t0, self.n, M=self.M, iterations=2, eff_impl=False, info=self._ba_info)
            except Exception as e:  # same recovery as the reference (:302-306)
                print(f"WARNING: BA failed...{e}")
            ixm = torch.arange(self.m, device=self.device) // self.M
            self.points_[:self.m] = pops.point_cloud(self.poses, self.patches_.view(-1, 3, 3, 3)[:self.m],
                                                     self.intrinsics, ixm)

    def _frame_stores_stepwise(self, n, slot, k_dev, kq, patches, imap, gmap, fmap, clr, ex):
        """the writes of reference Ramp_vo.py:346-381, one step at a time"""
        self.tstamps_[n].fill_(self.counter)
        self.index_map_[n + 1].fill_(self.m + self.M)
        if k_dev is None and n > 0:
            self.intrinsics_[n] = self.intrinsics_[n - 1]
        else:
            self.intrinsics_[n] = k_dev if k_dev is not None else self._upload(kq.astype(np.float32))
            self._last_K, self._last_K_raw = kq, self._K_raw_now
        if n > 1:
            if self.cfg.MOTION_MODEL == 'DAMPED_LINEAR':
                P1 = SE3(self.poses_[n - 1])
                P2 = SE3(self.poses_[n - 2])
                xi = self.cfg.MOTION_DAMPING * (P1 * P2.inv()).log()
                self.poses_[n] = (SE3.exp(xi) * P1).data
            else:
                self.poses_[n] = self.poses_[n - 1]
        if self.is_initialized:
            patches[:, :, 2] = torch.median(self.patches_[n - 3:n, :, 2])
        else:
            patches[:, :, 2] = self._initial_depth(patches)
        clr = (clr[0][:, [2, 1, 0]] + 0.5) * (255.0 / 2)
        self.colors_[n] = clr.to(torch.uint8)
        self.patches_[n] = patches
        self.imap_[slot] = imap.reshape(self.M, self.DIM).to(self.dtype)
        self.gmap_[slot] = gmap[0].permute(0, 2, 3, 1).to(self.dtype)
        f = fmap[0]
        self.fmap1_[slot] = f[0].permute(1, 2, 0).to(self.dtype)
        self.fmap2_[slot] = F.avg_pool2d(f, 4, 4)[0].permute(1, 2, 0).to(self.dtype)
