"""Network operators of the tracking loop (reference: ramp/net.py).

``VONet`` keeps the surface ``Ramp_vo`` consumes -- ``patchify(...)``,
``update(net, inp, corr, flow, ii, jj, kk)`` and the attributes ``DIM, RES, P`` --
and the reference's parameter names (checkpoints load with strict=True).  The
training unroll ``VONet.forward`` is outside the hot path and not provided.
"""
import torch
import torch.nn as nn

from . import altcorr, fastba, ops
from ._lib import RAMP_NCHW, RAMP_NHWC
from .blocks import GatedResidual, SoftAgg
from .extractor import MergerLSTMsceneEncoder, MultiScaleMergerDoubleNet
from ._lib import scratch_owner as _lib_scratch_owner
from .utils import coords_grid_with_index, get_channel_dim, get_coords_from_topk_events, preprocess_input

DIM = 384


class GradientClip(nn.Module):
    """identity in the forward pass (reference blocks.py:76-90 clips gradients only);
    kept so the Sequential indices -- and checkpoint keys -- match"""

    def forward(self, x):
        return x


class GraphPlan:
    """Per-graph index structures shared by the Update operator and BA: temporal
    neighbours and the two SoftAgg groupings.  Built on the device in one go
    (``build``); ``Ramp_vo`` rebuilds it only when the factor graph changes."""
    __slots__ = ("ix", "jx", "ix_raw", "jx_raw", "mask_ix", "mask_jx", "g_kk", "g_ij", "max_kk", "max_ij", "E",
                 "pair_mul", "kj")

    def tensors(self):
        """every device tensor of the plan (for stream bookkeeping)"""
        out = [t for t in (self.ix, self.jx, self.ix_raw, self.jx_raw, self.mask_ix, self.mask_jx)
               if isinstance(t, torch.Tensor)]
        if isinstance(self.kj, torch.Tensor):
            out.append(self.kj)
        for g in (self.g_kk, self.g_ij):
            out += [getattr(g, n) for n in ("order", "gid", "seg_start", "ukeys", "ngroups")
                    if isinstance(getattr(g, n, None), torch.Tensor)]
        return out

    @staticmethod
    def build(ii, jj, kk, kk_bound=0, jj_bound=0, max_kk=None, max_ij=None, kk_range=None, frame_range=None):
        """kk_range=(lo, hi) / frame_range=(lo, hi): tight half-open ranges of kk and of the frame
        indices in ii, jj when the caller knows them (the tracker does): the groupings then come
        from the counting group-by and the neighbours from the kk groups, no radix sort."""
        p = GraphPlan()
        p.mask_ix = p.mask_jx = None
        p.kj = None                    # factors in (kk, jj) order, when the counting group-by built the plan
        p.E = ii.shape[0]
        small = (kk_range is not None and frame_range is not None and max_kk is not None and max_ij is not None
                 and ii.is_cuda and hasattr(ops, "group_by_small"))
        if small:
            k_lo, k_hi = int(kk_range[0]), int(kk_range[1])
            f_lo, f_hi = int(frame_range[0]), int(frame_range[1])
            W = max(f_hi - f_lo, 1)
            p.g_kk = ops.group_by_small(kk, None, 1, k_lo, max(k_hi - k_lo, 1), max_kk)
            # (jj, ii) lexicographic: g_ij.order is target-frame-major, which is also the schedule
            # the correlation kernel wants (ops.corr(order=))
            p.g_ij = ops.group_by_small(jj, ii, W, f_lo * W + f_lo, W * W, max_ij)
            p.pair_mul = W                 # g_ij.ukeys = jj * W + ii
            p.ix, p.jx, p.kj = ops.neighbors_from_groups(p.g_kk, jj, max_kk, want_kj=True)
        else:
            p.ix, p.jx = ops.neighbors(kk, jj, kk_bound, jj_bound)
            p.g_kk = ops.group_by(kk, kk_bound)
            # keyed by (jj, ii) lexicographically -- the same partition as the reference's ii*12345+jj
            nb = int(jj_bound) if jj_bound else 0
            p.pair_mul = nb if nb else 12345
            p.g_ij = ops.group_by(jj * p.pair_mul + ii, nb * nb if nb else 0)
        p.ix_raw, p.jx_raw = p.ix, p.jx
        # group counts: caller-supplied upper bounds avoid a device->host read-back
        p.max_kk = int(max_kk) if max_kk is not None else int(p.g_kk.ngroups.item())
        p.max_ij = int(max_ij) if max_ij is not None else int(p.g_ij.ngroups.item())
        return p


class Update(nn.Module):
    """recurrent update operator; reference ramp/net.py:34-90"""

    def __init__(self, p):
        super().__init__()
        self.c1 = nn.Sequential(nn.Linear(DIM, DIM), nn.ReLU(inplace=True), nn.Linear(DIM, DIM))
        self.c2 = nn.Sequential(nn.Linear(DIM, DIM), nn.ReLU(inplace=True), nn.Linear(DIM, DIM))
        self.norm = nn.LayerNorm(DIM, eps=1e-3)
        self.agg_kk = SoftAgg(DIM)
        self.agg_ij = SoftAgg(DIM)
        self.gru = nn.Sequential(nn.LayerNorm(DIM, eps=1e-3), GatedResidual(DIM),
                                 nn.LayerNorm(DIM, eps=1e-3), GatedResidual(DIM))
        self.corr = nn.Sequential(nn.Linear(2 * 49 * p * p, DIM), nn.ReLU(inplace=True), nn.Linear(DIM, DIM),
                                  nn.LayerNorm(DIM, eps=1e-3), nn.ReLU(inplace=True), nn.Linear(DIM, DIM))
        self.d = nn.Sequential(nn.ReLU(inplace=False), nn.Linear(DIM, 2), GradientClip())
        self.w = nn.Sequential(nn.ReLU(inplace=False), nn.Linear(DIM, 2), GradientClip(), nn.Sigmoid())

    def fused(self, dtype):
        """row-fused HIP implementation (csrc/update.hip), one instance per GEMM dtype"""
        from .update_fused import FusedUpdate
        if not hasattr(self, "_fused_impl"):
            object.__setattr__(self, "_fused_impl", {})
        if dtype not in self._fused_impl:
            self._fused_impl[dtype] = FusedUpdate(self, dtype)
        return self._fused_impl[dtype]

    def forward(self, net, inp, corr, flow, ii, jj, kk, plan=None):
        from ._lib import require_cuda
        require_cuda(net, inp, corr)
        if plan is None:
            plan = GraphPlan.build(ii, jj, kk)
        # GEMMs + row-fused glue kernels / fused MFMA chains (update_fused.py); dtype = the caller's feature dtype
        fu = self.fused(corr.dtype)
        out32, relu_t = fu.hidden(net[0].float().contiguous(), inp[0].to(corr.dtype).contiguous(), None, 0,
                                  corr[0].contiguous(), plan)
        hw = fu.heads(relu_t)
        return out32[None], (hw[None, :, :2], torch.sigmoid(hw[None, :, 2:]), None)


class Patchifier(nn.Module):
    """encoder + patch selection + patch extraction; reference ramp/net.py:93-203"""

    def __init__(self, channels_dim, patch_size=3, input_mode="MultiScale"):
        super().__init__()
        self.input_mode = input_mode
        self.P = patch_size
        evs_ch_dim, img_ch_dim = channels_dim
        if input_mode == "SingleScale":
            self.encoder = MergerLSTMsceneEncoder(evs_ch_dim=evs_ch_dim, img_ch_dim=img_ch_dim, output_lstm_dim=15,
                                                  output_dim_f=128, output_dim_i=DIM, norm_fn_fmap="instance",
                                                  norm_fn_imap="none", kernel_size_superstate=1)
        elif input_mode == "MultiScale":
            self.encoder = MultiScaleMergerDoubleNet(evs_ch_dim=evs_ch_dim, img_ch_dim=img_ch_dim, lstm_dim=16,
                                                     output_dim_f=128, output_dim_i=DIM, norm_fn_fmap="instance",
                                                     norm_fn_imap="none", norm_superstate=False)
        else:
            raise ValueError(f"Invalid input mode: {input_mode}")
        self._grid = None
        import os
        self.use_graph = os.environ.get("RAMP_NO_GRAPH", "0") != "1"
        self._graphs = {}
        self._graph_warm = 0
        # fp32 features: the tracker's pyramid planes chunked for the MFMA correlation kernels -- 2 (default): split fp16
        # parts [h][4][2][w][32] for corr_mfma_kernel<CorrX2>; 1: [h][8][w][16] fp32 for corr_mfma_kernel<float>; 0
        # (RAMP_CORR_F32_MFMA=0): plain NHWC planes for corr_kernel<float>, the reference kernel's summation order
        from ._lib import corr_f32_mode
        self.pack_f32 = corr_f32_mode()
        self._plist = None
        self._extra = None
        self._index = None

    def _apply(self, fn, *a, **k):
        """.to() / .half() / .cuda() replace the parameter tensors: captured graphs point at the old ones"""
        self._graphs, self._graph_warm, self._plist = {}, 0, None
        return super()._apply(fn, *a, **k)

    def _coord_grid(self, h, w, device):
        if self._grid is None or self._grid.shape[-2:] != (h, w) or self._grid.device != device:
            disps = torch.ones(1, 1, h, w, device=device)
            grid, _ = coords_grid_with_index(disps, device=device)
            self._grid = grid[0].contiguous()       # [1,3,h,w]
        return self._grid

    def forward(self, input_, patches_per_image=80, reinit_hidden=False, disps=None, event_bias=False,
                gradient_bias=False, pre_replay=None):
        """On the GPU the whole front-end (fused LSTM, ~35 conv-tower launches, patch
        selection, 4 patch gathers: ~60 launches of static shape) is captured into ONE hipGraph
        after a warm-up call and replayed per frame; results live in the graph's static output
        buffers until the next call (Ramp_vo copies them into its ring buffers immediately)."""
        events, images, mask = input_
        # pre_replay (optional callable): run once everything that depends only on the INPUTS has been enqueued -- the
        # staging copies and the patch selection (a function of the events alone) -- and before the encoder: the
        # device-resident tracker gates the encoder behind the previous frame's update operator there, while the
        # staging and the selection (40 us of small launches) run ahead of the gate
        graphable = (self.use_graph and events.is_cuda and event_bias
                     and disps is None and not reinit_hidden and events.shape[1] == 1 and not torch.is_grad_enabled())
        if graphable and self.input_mode != "SingleScale":
            # MultiScale: only frames that are present (host-side mask) share one static graph
            graphable = mask is not None and mask.device.type == "cpu" and mask.numel() == 1 and bool(mask.all())
        if not graphable:
            self._graph_warm = 0 if reinit_hidden else self._graph_warm
            if pre_replay is not None:
                pre_replay()
            return self._forward_impl(input_, patches_per_image, reinit_hidden, disps, event_bias, gradient_bias)
        # (the patch selection as a forked branch of the graph -- beside the recurrent encoder step instead of ahead of it -- was
        # measured in round 5: front end alone 259 -> 277 us, sequential rate 937 -> 932 kf/s; not kept)
        key = (self.input_mode, tuple(events.shape), tuple(images.shape), patches_per_image, events.dtype,
               images.dtype, bool(getattr(self.encoder, "mixed_precision", False)),
               bool(getattr(self.encoder, "fp8_mfma", False)), events.device)
        # the captured graph bakes in raw pointers to the packed encoder weights and to the encoder's recurrent
        # state buffers: it is only valid for the parameter values and the state object it was captured with
        # (in-place weight updates bump ``_version``; a resolution change reallocates the state)
        if self._plist is None:
            self._plist = list(self.encoder.parameters())
        sig = (sum(p._version for p in self._plist), self._plist[0].data_ptr(), id(getattr(self.encoder, "_hip_state", None)))
        g = self._graphs.get(key)
        if g is not None and g[5] != sig:
            del self._graphs[key]              # stale: captured against other weights / another state buffer
            g, self._graph_warm = None, 0
        if g is None:
            if pre_replay is not None:
                pre_replay()
            if self._graph_warm < 1:      # one eager call with carried state first (allocator / pack caches warm)
                self._graph_warm += 1
                return self._forward_impl(input_, patches_per_image, False, None, event_bias, gradient_bias)
            ev_s, im_s = events.clone(), images.clone()
            # the patch centres are a function of the events alone: selected OUTSIDE the graph into a static buffer
            # (three small launches that need not wait for the encoder's turn, see pre_replay)
            coords_s = self._select(ev_s, mask, patches_per_image, None)
            graph = torch.cuda.CUDAGraph()
            # (no cyclic collection while the capture runs: freeing another tracker's hipGraph -- an object the collector
            # may find at any allocation -- inside a capture aborts the process)
            import gc
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(graph):
                    outs = self._forward_impl((ev_s, im_s, mask), patches_per_image, False, None, event_bias,
                                              gradient_bias, coords_in=coords_s)
            finally:
                if gc_was_on:
                    gc.enable()
            sig = (sig[0], sig[1], id(getattr(self.encoder, "_hip_state", None)))
            self._graphs[key] = g = (graph, ev_s, im_s, outs, self._extra, sig, coords_s)
            self._run_graph(graph)       # capture does not execute: run this frame now (inputs already staged)
            return outs
        graph, ev_s, im_s, outs, self._extra, _, coords_s = g
        if (events.is_contiguous() and images.is_contiguous() and events.dtype == ev_s.dtype and images.dtype == im_s.dtype
                and (events.numel() * events.element_size()) % 16 == 0 and (images.numel() * images.element_size()) % 16 == 0
                and events.data_ptr() % 16 == 0 and images.data_ptr() % 16 == 0):
            ops.multi_copy([(events, ev_s), (images, im_s)])          # one launch for the two staging copies
        else:
            ev_s.copy_(events)
            im_s.copy_(images)
        if coords_s is not None:
            self._select(ev_s, mask, patches_per_image, coords_s)
        if pre_replay is not None:
            pre_replay()
        self._run_graph(graph)
        return outs

    def _run_graph(self, graph):
        graph.replay()

    def _select(self, events, mask, patches_per_image, out):
        """patch centres [1, M, 2] of the frame(s) present (reference net.py:176-180), into ``out`` when given"""
        if self.input_mode != "SingleScale" and mask is not None and not bool(mask.all()):
            events = events[mask]
        with _lib_scratch_owner(self):
            c = get_coords_from_topk_events(events=events, patches_per_image=patches_per_image,
                                            border_suppression_size=0, non_max_supp_rad=11,
                                            out=out[0] if out is not None else None)
        return c.float().contiguous() if out is None else out

    def _forward_impl(self, *a, **k):
        from . import _lib
        with _lib.scratch_owner(self):          # (captured graphs must not share scratch: _lib.workspace)
            return self._forward_steps(*a, **k)

    def _forward_steps(self, input_, patches_per_image=80, reinit_hidden=False, disps=None, event_bias=False,
                       gradient_bias=False, coords_in=None):
        events, images, mask = input_
        if self.input_mode == "SingleScale":
            fmap, imap, _ = self.encoder(events=events, images=images, reinit_hidden=reinit_hidden,
                                         out_scale=0.25)           # fmap / 4.0, imap / 4.0 folded in
            mask = None
        else:
            fmap, imap = self.encoder(events=events, images=images, mask=mask, reinit_hidden=reinit_hidden,
                                      out_scale=0.25)
            if not bool(mask.all()):
                events = events[mask]
        if mask is not None and not mask.any():
            return None, None, None, None, None, None
        b, n, c, h, w = fmap.shape
        if coords_in is not None:
            coords = coords_in           # selected ahead of the captured graph (forward())
        elif event_bias:
            coords = get_coords_from_topk_events(events=events, patches_per_image=patches_per_image,
                                                 border_suppression_size=0, non_max_supp_rad=11)
        else:
            if gradient_bias:
                raise NotImplementedError("gradient-biased sampling is a training-time option")
            x = torch.randint(1, w - 1, size=[n, patches_per_image], device=fmap.device)
            y = torch.randint(1, h - 1, size=[n, patches_per_image], device=fmap.device)
            coords = torch.stack([x, y], dim=-1).float()
        coords = coords.float().contiguous()
        # channels-last storage -> NHWC kernels; results are handed out in the reference's shapes
        f_nhwc = fmap[0].permute(0, 2, 3, 1)
        i_nhwc = imap[0].permute(0, 2, 3, 1)
        if (fmap.is_cuda and n == 1 and disps is None and f_nhwc[0].is_contiguous() and i_nhwc[0].is_contiguous()
                and images.dtype == torch.float32):
            # the tracking path: the four gathers and the steps around them as ONE launch
            g, ip, pt, cl, col = ops.frame_gather(f_nhwc[0], i_nhwc[0], images[0, 0], coords[0])
            gmap = g[None].permute(0, 1, 4, 2, 3)                             # [1,M,128,3,3] view
            imap_p = ip.view(b, -1, DIM, 1, 1)
            patches = pt.view(b, -1, 3, self.P, self.P)
            clr = cl.view(b, -1, 3)
            if self._index is None or self._index.shape[0] != patches_per_image or self._index.device != fmap.device:
                self._index = torch.zeros(patches_per_image, dtype=torch.long, device=fmap.device)
            chunked = ((fmap.dtype == torch.float16 or (fmap.dtype == torch.float32 and self.pack_f32))
                       and ops.pyramid_pack_supported(h, w))
            if chunked:
                f1, f2 = ops.pyramid_pack(f_nhwc[0], split=int(self.pack_f32) == 2)
            else:
                import torch.nn.functional as F
                f1 = f_nhwc[0]
                f2 = F.avg_pool2d(fmap[0], 4, 4).permute(0, 2, 3, 1).contiguous()[0]
            self._extra = dict(gmap=g[None], imap=ip, fmap=f1, fmap2=f2, colors=col, chunked=chunked)
            return fmap, gmap, imap_p, patches, self._index, clr
        gmap = ops.patchify(f_nhwc, coords, 1, True, RAMP_NHWC, RAMP_NHWC)           # [n,M,3,3,128]
        gmap = gmap.permute(0, 1, 4, 2, 3).reshape(b, -1, 128, self.P, self.P)   # a view when n == 1
        imap_p = ops.patchify(i_nhwc, coords, 0, True, RAMP_NHWC, RAMP_NHWC)         # [n,M,1,1,384]
        imap_p = imap_p.view(b, -1, DIM, 1, 1)
        if disps is None:
            grid = self._coord_grid(h, w, fmap.device).expand(n, 3, h, w).contiguous()
        else:
            grid, _ = coords_grid_with_index(disps, device=fmap.device)
            grid = grid[0].contiguous()
        patches = altcorr.patchify(grid, coords, self.P // 2).view(b, -1, 3, self.P, self.P)
        index = torch.arange(n, device=fmap.device).view(n, 1).repeat(1, patches_per_image).reshape(-1)
        clr = altcorr.patchify(images[0].float(), 4 * (coords + 0.5), 0).view(b, -1, 3)
        if fmap.is_cuda and n == 1:
            # extras for the tracker's one-launch state store (channels-last sources, the 1/4 pyramid
            # level and the uint8 BGR colours of Ramp_vo.py:353-354, 381), produced inside the graph
            col = ((clr[0].flip(-1) + 0.5) * (255.0 / 2)).to(torch.uint8)       # BGR, no host index tensor
            chunked = ((fmap.dtype == torch.float16 or (fmap.dtype == torch.float32 and self.pack_f32))
                       and ops.pyramid_pack_supported(h, w) and f_nhwc[0].is_contiguous())
            if chunked:
                # both correlation levels in the MFMA kernel's [h][C/32][w][32] target layout
                f1, f2 = ops.pyramid_pack(f_nhwc[0], split=int(self.pack_f32) == 2)
            else:
                import torch.nn.functional as F
                f1 = f_nhwc[0]
                f2 = F.avg_pool2d(fmap[0], 4, 4).permute(0, 2, 3, 1).contiguous()[0]
            self._extra = dict(gmap=gmap.permute(0, 1, 3, 4, 2), imap=imap_p.view(-1, DIM), fmap=f1,
                               fmap2=f2, colors=col, chunked=chunked)
        else:
            self._extra = None
        return fmap, gmap, imap_p, patches, index, clr


class VONet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.P = 3
        self.RES = 4
        self.DIM = DIM
        self.EVENT_BIAS = cfg["event_bias"]
        self.MOTION_MODEL = "DAMPED_LINEAR"
        self.MOTION_DAMPING = 0.5
        self.inp_channel_dims = get_channel_dim(cfg)
        self.input_mode = cfg["input_mode"]
        self.patchify = Patchifier(channels_dim=self.inp_channel_dims, patch_size=self.P, input_mode=self.input_mode)
        self.update = Update(self.P)

    def forward(self, *args, **kwargs):
        raise NotImplementedError("the training unroll (reference net.py:252-378) is outside the tracking hot "
                                  "path; use Ramp_vo for inference")
