"""Host-side helpers of the tracking loop (reference: ramp/utils.py, the seven
functions on the hot path; viz / losses / IO are out of scope)."""
import torch
import torch.nn.functional as F


class Timer:
    """reference utils.py:22-43 (event pair around a region, prints ms)"""
    all_times = []

    def __init__(self, name, enabled=True):
        self.name, self.enabled = name, enabled
        if enabled:
            self.start = torch.cuda.Event(enable_timing=True)
            self.end = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.enabled:
            self.start.record()

    def __exit__(self, *exc):
        if self.enabled:
            self.end.record()
            torch.cuda.synchronize()
            ms = self.start.elapsed_time(self.end)
            Timer.all_times.append(ms)
            print(self.name, ms)


def flatmeshgrid(*args, **kwargs):
    return (x.reshape(-1) for x in torch.meshgrid(*args, **kwargs))


def coords_grid_with_index(d, **kwargs):
    """(x, y, disparity) grid [b,n,3,h,w] and frame-index grid (reference utils.py:54-69)"""
    b, n, h, w = d.shape
    x = torch.arange(0, w, dtype=torch.float, **kwargs)
    y = torch.arange(0, h, dtype=torch.float, **kwargs)
    yy, xx = torch.meshgrid(y, x, indexing="ij")
    yy = yy.view(1, 1, h, w).expand(b, n, h, w)
    xx = xx.view(1, 1, h, w).expand(b, n, h, w)
    coords = torch.stack([xx, yy, d], dim=2)
    index = torch.arange(0, n, dtype=torch.float, **kwargs).view(1, n, 1, 1, 1).expand(b, n, 1, h, w)
    return coords, index


def nms_image(x, kernel_size=3):
    """keep local maxima of each channel of x [C,H,W] (reference utils.py:157-183)"""
    pad = (kernel_size - 1) // 2
    mx = F.max_pool2d(x.unsqueeze(0), kernel_size, stride=1, padding=pad).squeeze(0)
    return x * (mx == x).float()


def get_coords_from_topk_events(events, patches_per_image, border_suppression_size=0,
                                non_max_supp_rad=0, out=None):
    """patch centres at the top-k cells of the NMS'ed mean |event| map at 1/4
    resolution (reference utils.py:186-226).  The map is laid out [w, h] and the
    reference derives x by TRUE division of the flat index by h, so x carries the
    fraction y/h -- reproduced because every later bilinear lookup depends on it."""
    e4 = events.squeeze(0)                                        # [T, bins, H, W]
    if e4.is_cuda and e4.shape[0] == 1 and border_suppression_size == 0:
        from . import ops
        if ops.event_topk_supported(e4[0], patches_per_image, non_max_supp_rad):
            # GPU: score + NMS + radix select in three HIP launches (csrc/select.hip)
            return ops.event_topk(e4[0], patches_per_image, non_max_supp_rad, out=out)[None]
    assert out is None
    ev = torch.abs(e4)
    ev = F.avg_pool2d(ev, 4, 4).transpose(3, 2).mean(dim=1)      # [T, w, h]
    if border_suppression_size != 0:
        b = border_suppression_size
        ev[:, :b, :] = 0
        ev[:, -b:, :] = 0
        ev[:, :, :b] = 0
        ev[:, :, -b:] = 0
    if non_max_supp_rad != 0:
        ev = nms_image(ev, kernel_size=non_max_supp_rad)
    hh = ev.shape[-1]
    flat = torch.flatten(ev, start_dim=1)
    _, indices = torch.topk(flat, k=patches_per_image, dim=-1)
    rows = indices / hh           # float: x + y/h
    cols = indices % hh
    return torch.stack((rows, cols.to(rows.dtype)), dim=-1)


def check_input_tensors(events, images):
    if len(events.shape) != len(images.shape):
        raise AssertionError("Event and image tensor must have the same number of dimension")
    if len(events.shape) != 5:
        raise AssertionError("Event and image tensor must have shape [batch, n_tensors, channels, height, width]")
    if not (events.shape[0] == 1 and images.shape[0] == 1):
        raise NotImplementedError("Event and image tensor must have batch dimension (0 dim) = 1. "
                                  "Multiple batches not yet implemented")


def get_channel_dim(cfg):
    return (cfg["num_event_bins"], 3)


def preprocess_input(input_tensor):
    mask = None
    if len(input_tensor) < 3:
        events, images = input_tensor
    else:
        events, images, mask = input_tensor
    check_input_tensors(events, images)
    return (events, images, mask)


def filter_features(confidences, target, data_shape):
    """zero the confidence of targets outside [0,wd]x[0,ht] (reference utils.py:557-570)"""
    ht, wd = data_shape
    x, y = target[..., 0], target[..., 1]
    outside = (x < 0) | (x > wd) | (y < 0) | (y > ht)
    return confidences * (~outside).unsqueeze(-1).to(confidences.dtype)
