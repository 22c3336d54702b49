"""RAMP encoders (reference: ramp/extractor.py).

The module tree and parameter names mirror the reference so its checkpoints load
with ``strict=True`` (104 keys SingleScale / 116 MultiScale); the forward passes
are this package's own:

  * everything is channels-last: a pixel's channels are contiguous, so the
    per-pixel recurrent cells are plain [H*W, C] row-major GEMV batches and the
    conv towers feed the correlation kernels without a layout change;
  * the per-pixel LSTM + super-state 1x1 convolutions are evaluated as fused
    pointwise math on [H*W, C] matrices (the reference calls cuDNN's LSTM on
    307,200 length-1 sequences and syncs on ``torch.any`` per modality);
  * the conv towers are the HIP implicit-GEMM MFMA kernels of csrc/conv.hip (conv_hip.py).

There is no ATen path in this package: the modules hold the parameters (and load the reference's checkpoints);
their forward runs the HIP front end or raises.  The plain-PyTorch restatement used as the numerics reference and
by the CPU baseline lives with the test infrastructure (oracle/host_cpu.py).
"""
import torch
import torch.nn as nn

from ._lib import require_cuda

DIM = 32


def _norm(kind, planes):
    if kind == 'instance':
        return nn.InstanceNorm2d(planes)
    if kind == 'none':
        return nn.Sequential()
    if kind == 'batch':
        return nn.BatchNorm2d(planes)
    if kind == 'group':
        return nn.GroupNorm(num_groups=planes // 8, num_channels=planes)
    raise ValueError(kind)


class ResidualBlock(nn.Module):
    """two 3x3 convs + skip (1x1 strided conv when downsampling); reference :8-57"""

    def __init__(self, in_planes, planes, norm_fn='group', stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm_fn = norm_fn
        self.norm1 = _norm(norm_fn, planes)
        self.norm2 = _norm(norm_fn, planes)
        self.downsample = None
        if stride != 1:
            self.norm3 = _norm(norm_fn, planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def forward(self, x):
        raise RuntimeError("rampvo_amd: the conv towers run as a whole on the HIP kernels (conv_hip.basic_encoder4 / "
                           "multiscale_encoder4); there is no per-module ATen forward")


class BasicEncoder4(nn.Module):
    """conv7x7/2 -> 2xRes(32) -> Res(32->64,/2)+Res(64) -> conv1x1; reference :60-130"""

    def __init__(self, output_dim=128, norm_fn='batch', dropout=0.0, multidim=False, channel_dim=5):
        super().__init__()
        self.norm_fn = norm_fn
        self.channel_dim = channel_dim
        self.norm1 = _norm(norm_fn, DIM)
        self.conv1 = nn.Conv2d(channel_dim, DIM, kernel_size=7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.in_planes = DIM
        self.layer1 = self._make_layer(DIM, stride=1)
        self.in_planes_layer1 = self.in_planes
        self.layer2 = self._make_layer(2 * DIM, stride=2)
        self.conv2 = nn.Conv2d(2 * DIM, output_dim, kernel_size=1)
        self.dropout = None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _make_layer(self, output_dim, stride=1):
        layers = (ResidualBlock(self.in_planes, output_dim, self.norm_fn, stride=stride),
                  ResidualBlock(output_dim, output_dim, self.norm_fn, stride=1))
        self.in_planes = output_dim
        return nn.Sequential(*layers)

    def forward(self, x, out_scale=1.0):
        """x [b,n,c,h,w] -> [b,n,out,h/4,w/4]: runs inside the encoder's HIP front end only"""
        raise RuntimeError("rampvo_amd: the conv towers run as a whole on the HIP kernels (conv_hip.basic_encoder4)")


class MultiScaleBasicEncoder4(BasicEncoder4):
    """three-scale tower: the 1/2 and 1/4 super-states are concatenated in;
    reference :274-311.  ``layer2`` / ``conv2`` exist (and are loaded) but unused,
    as upstream."""

    def __init__(self, output_dim=128, internal_input_dimensions=None, **kwargs):
        super().__init__(**kwargs)
        if internal_input_dimensions is None:
            internal_input_dimensions = [self.channel_dim] * 3
        self.internal_input_dimensions = internal_input_dimensions
        self.in_planes = self.in_planes_layer1 + internal_input_dimensions[1]
        self.layer3 = self._make_layer(output_dim=2 * DIM, stride=2)
        self.conv3 = nn.Conv2d(2 * DIM + internal_input_dimensions[2], output_dim, kernel_size=1)

    def forward(self, x, x_down2, x_down4, out_scale=1.0):
        raise RuntimeError("rampvo_amd: the conv towers run as a whole on the HIP kernels (conv_hip.multiscale_encoder4)")


class MergerLSTMsceneEncoder(nn.Module):
    """SingleScale RAMP encoder; reference :187-269"""

    def __init__(self, evs_ch_dim=5, img_ch_dim=3, output_lstm_dim=15, output_dim_f=128, output_dim_i=DIM,
                 norm_fn_fmap="instance", norm_fn_imap="none", kernel_size_superstate=1):
        super().__init__()
        assert kernel_size_superstate == 1
        self.hidden_size = output_lstm_dim
        self.events_convlstm = nn.LSTM(input_size=evs_ch_dim, hidden_size=output_lstm_dim, batch_first=True)
        self.image_convlstm = nn.LSTM(input_size=img_ch_dim, hidden_size=output_lstm_dim, batch_first=True)
        self.superstate_encoder = nn.Conv2d(2 * output_lstm_dim, output_lstm_dim, kernel_size=1)
        self.fmap_encoder = BasicEncoder4(output_dim=output_dim_f, norm_fn=norm_fn_fmap, channel_dim=output_lstm_dim)
        self.imap_encoder = BasicEncoder4(output_dim=output_dim_i, norm_fn=norm_fn_imap, channel_dim=output_lstm_dim)
        self.states_events, self.states_image, self.super_state = None, None, None   # (oracle-side torch forward)
        self._hip_state = None
        self.mixed_precision = False      # fp16 storage / fp16 MFMA conv towers (set by Ramp_vo from cfg)
        self.fp8_mfma = False             # with mixed_precision: the towers' products on the fp8 MFMA (configs[4])

    def _forward_hip(self, events, images, reinit_hidden, out_scale):
        """fused LSTM/super-state kernel + MFMA conv towers (csrc/conv.hip)"""
        from . import conv_hip
        H, W = events.shape[-2:]
        st = self._hip_state
        if st is None or st.HW != H * W or st.ss.device != events.device:
            st = self._hip_state = conv_hip.LstmState(H * W, events.device)
        if reinit_hidden:
            st.fresh = True
        s16 = conv_hip.lstm_superstate_step(self, events[0, 0].float().contiguous(),
                                            images[0, 0].float().contiguous(), st, arena_for_towers=self.mixed_precision)
        # both towers layer by layer; fp16: one launch per layer for the two of them ([h,w,128], [h,w,384]).  One stream:
        # a fork / join inside the front end's hipGraph bought nothing measurable (the paired launches fill the
        # chip) and multi-stream captures were the one configuration that crashed hipGraphLaunch in long test runs.
        f, i = conv_hip.basic_encoder4_towers([self.fmap_encoder, self.imap_encoder], s16, out_scale,
                                              half=self.mixed_precision, fp8=self.mixed_precision and self.fp8_mfma)
        return f.permute(2, 0, 1)[None, None], i.permute(2, 0, 1)[None, None], None

    def forward(self, events, images, reinit_hidden=False, out_scale=1.0):
        B, T, Ce, H, W = events.shape
        require_cuda(events, images)
        if B != 1 or T != 1 or images.shape[1] != 1:
            raise RuntimeError("rampvo_amd: the encoder advances one frame per call (batch 1, T = 1: the tracking path)")
        return self._forward_hip(events, images, reinit_hidden, out_scale)


class LSTMEncoder(nn.Module):
    """strided conv + per-pixel LSTM with NO state carry; reference :314-390"""

    def __init__(self, in_channels, downsample_scale=0, out_channels=15, batch_norm_momentum=0.1,
                 activation_fn=None, normalization_type=None):
        super().__init__()
        assert activation_fn is None and normalization_type is None
        k, stride, pad = downsample_scale + 1, downsample_scale, 1
        if downsample_scale <= 1:
            k, stride, pad = 1, 1, 0
        self.conv_1 = nn.Conv2d(in_channels, in_channels, kernel_size=k, stride=stride, padding=pad)
        self.convlstm = nn.LSTM(input_size=in_channels, hidden_size=out_channels, batch_first=True)
        self.norm_layer = nn.Sequential()

    def forward(self, x):
        raise RuntimeError("rampvo_amd: fused into ms_lstm_superstate_kernel (conv_hip.ms_lstm_superstate_step)")


class SuperStateEncoder(nn.Module):
    """1x1 conv over [state ; embedding]; reference :393-463"""

    def __init__(self, kernel_size, out_channels=15, norm_superstate=False):
        super().__init__()
        assert kernel_size == 1 and not norm_superstate
        self.encoder = nn.Conv2d(2 * out_channels, out_channels, kernel_size=1)
        self.instance_norm_layer = nn.InstanceNorm2d(num_features=out_channels)
        self.norm_superstate = norm_superstate


class MultiScaleMergerDoubleNet(nn.Module):
    """MultiScale RAMP encoder; reference :468-566"""

    def __init__(self, evs_ch_dim, img_ch_dim, lstm_dim=16, output_dim_f=128, output_dim_i=DIM,
                 norm_fn_fmap="instance", norm_fn_imap="none", kernel_size_superstate=1, activation_fn=None,
                 normalization_type=None, norm_superstate=False):
        super().__init__()
        self.scales = [1, 2, 4]
        self.ev_encoders, self.im_encoders = nn.ModuleList(), nn.ModuleList()
        self.super_state_ev_encoder, self.super_state_im_encoders = nn.ModuleList(), nn.ModuleList()
        self.internal_dimensions = []
        self.super_states = []
        for s in self.scales:
            d = lstm_dim * s
            self.internal_dimensions.append(d)
            self.ev_encoders.append(LSTMEncoder(evs_ch_dim, downsample_scale=s, out_channels=d))
            self.im_encoders.append(LSTMEncoder(img_ch_dim, downsample_scale=s, out_channels=d))
            self.super_state_ev_encoder.append(SuperStateEncoder(kernel_size_superstate, d, norm_superstate))
            self.super_state_im_encoders.append(SuperStateEncoder(kernel_size_superstate, d, norm_superstate))
            self.super_states.append(None)
        self.fmap_encoder = MultiScaleBasicEncoder4(output_dim=output_dim_f, norm_fn=norm_fn_fmap,
                                                    channel_dim=lstm_dim,
                                                    internal_input_dimensions=self.internal_dimensions)
        self.imap_encoder = MultiScaleBasicEncoder4(output_dim=output_dim_i, norm_fn=norm_fn_imap,
                                                    channel_dim=lstm_dim,
                                                    internal_input_dimensions=self.internal_dimensions)
        self._hip_state = None
        self.mixed_precision = False      # fp16 storage / fp16 MFMA conv towers (set by Ramp_vo from cfg)
        self.fp8_mfma = False             # with mixed_precision: the towers' products on the fp8 MFMA (configs[4])

    def _forward_hip(self, events, images, present, reinit_hidden, out_scale):
        """one time step on the GPU: fused conv_1 + LSTM + super-state kernel per scale, then the MFMA
        conv towers (csrc/conv.hip).  ``present``: the frame's mask (events-only steps advance the
        super-states and produce no maps)."""
        from . import conv_hip
        H, W = events.shape[-2:]
        ev, im = events[0, 0].float().contiguous(), images[0, 0].float().contiguous()
        st = self._hip_state
        if st is None or st[0].s.device != events.device or (st[0].Hs, st[0].Ws) != (H, W):
            st = self._hip_state = [conv_hip.MsState(H, W, s, events.device) for s in self.scales]
        xs = []
        for k in range(len(self.scales)):
            if reinit_hidden:
                st[k].fresh = True
            # (fp16 towers: scales 2 and 4 enter them as half tensors -- the kernel writes that copy itself)
            xs.append(conv_hip.ms_lstm_superstate_step(self, k, ev, im, st[k], present,
                                                       want_half=self.mixed_precision and k > 0))
        if not present:
            return None, None
        half = self.mixed_precision
        f, i = conv_hip.multiscale_encoder4_towers([self.fmap_encoder, self.imap_encoder], xs[0], xs[1], xs[2],
                                                   out_scale, half=half, fp8=half and self.fp8_mfma)
        return f.permute(2, 0, 1)[None, None], i.permute(2, 0, 1)[None, None]

    def forward(self, events, images, mask, reinit_hidden=False, out_scale=1.0):
        mask_list = [bool(m) for m in mask.reshape(-1).tolist()]
        require_cuda(events, images)
        if events.shape[0] != 1 or events.shape[1] != 1 or len(mask_list) != 1:
            raise RuntimeError("rampvo_amd: the encoder advances one frame per call (batch 1, T = 1: the tracking path)")
        return self._forward_hip(events, images, mask_list[0], reinit_hidden, out_scale)
