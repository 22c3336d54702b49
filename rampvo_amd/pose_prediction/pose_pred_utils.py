"""Helpers of the pose-prediction mode (reference: ramp/pose_prediction/pose_pred_utils.py).

The mode extrapolates a *virtual* keyframe: every live patch gets one more factor into a frame that has no
image yet; the patch's past reprojections (one per existing factor) are fitted with a weighted smoothing
spline per coordinate, the spline is evaluated ``step`` frames ahead, and two bundle-adjustment iterations
move the virtual pose (started from the damped-linear motion model) towards those targets.

Same functions, argument meaning and results as the reference's helpers; what differs is how they run:
the reference walks the factor list in Python with one device->host read and three full-length masks per
patch (``compute_patch_track__`` :172-189, ``predict_patch_on_model`` :325-345: O(patches x E) device work);
here tracks are cut out of the factor list with one stable sort and written back with one indexed store.
The spline itself is scipy's FITPACK wrapper on the host, as upstream (a handful of 5-point fits per patch).
Plotting helpers (:60-120) are not provided.
"""
from collections import OrderedDict

import numpy as np
import torch
from scipy.interpolate import UnivariateSpline
from scipy.spatial.transform import Rotation

from ..lietorch import SE3

PAST_PATCH_NUM = 5          # samples per fit (reference :303)


def compute_relative_pose_error(pose1, pose2):
    """(translation error, rotation error in degrees) between two [x y z qx qy qz qw] poses -- reference :13-49
    (including its quaternion re-ordering before ``Rotation.from_quat``)"""
    pose1 = np.array(pose1, dtype=float)
    pose2 = np.array(pose2, dtype=float)
    t1, t2 = pose1[:3], pose2[:3]
    q1 = pose1[3:] / np.linalg.norm(pose1[3:])
    q2 = pose2[3:] / np.linalg.norm(pose2[3:])
    R1 = Rotation.from_quat([q1[3], q1[0], q1[1], q1[2]]).as_matrix()
    R2 = Rotation.from_quat([q2[3], q2[0], q2[1], q2[2]]).as_matrix()
    translation_error = ((t2 - t1) ** 2).sum() ** 0.5
    rotation_error = Rotation.from_matrix(R1.T @ R2).as_euler('xyz', degrees=True)
    rotation_error = (rotation_error ** 2).sum() ** 0.5
    return translation_error, rotation_error


def relative_pose_error(pose1, pose2):
    """|| Log(P1^-1 P2) ||  (reference :53-57; the upstream body calls ``np.ndarray(pose)``, which builds an
    uninitialised array of that SHAPE -- this is what the name and docstring say instead)"""
    p1 = torch.as_tensor(np.asarray(pose1), dtype=torch.float32)
    p2 = torch.as_tensor(np.asarray(pose2), dtype=torch.float32)
    return (SE3(p1).inv() * SE3(p2)).log().norm().item()


def motion_bootstrap(n, poses, MOTION_MODEL, MOTION_DAMPING):
    """pose of frame n from frames n-1, n-2 (reference :192-201)"""
    if MOTION_MODEL == 'DAMPED_LINEAR':
        P1 = SE3(poses[n - 1])
        P2 = SE3(poses[n - 2])
        xi = MOTION_DAMPING * (P1 * P2.inv()).log()
        return (SE3.exp(xi) * P1).data
    return poses[n - 1]


def add_forward_elements(frame_num, patch_extracted_num, r, ii, jj, kk, ix, weights):
    """append one factor per live patch into frame index frame_num-1 (reference :204-217)"""
    dev = ii.device
    t0 = patch_extracted_num * max(frame_num - r, 0)
    t1 = patch_extracted_num * max(frame_num - 1, 0)
    kk_toadd = torch.arange(t0, t1, device=dev)
    jj_toadd = torch.full_like(kk_toadd, frame_num - 1)
    ii_stack = torch.cat([ii, ix[kk_toadd]])
    jj_stack = torch.cat([jj, jj_toadd])
    kk_stack = torch.cat([kk, kk_toadd])
    new_weights = torch.zeros((1, len(kk_toadd), 2), device=dev, dtype=weights.dtype)
    return ii_stack, jj_stack, kk_stack, torch.cat([weights, new_weights], dim=1)


def compute_patch_track__(coords, ii, jj, kk, image_to_proj):
    """{(source frame, patch id): [L, 2] reprojections of the patch's pixel (0, 0), in factor-list order} for
    every patch that has a factor into frame ``image_to_proj`` (reference :172-189)"""
    into = jj == image_to_proj
    heads_k = kk[into]
    if heads_k.numel() == 0:
        return OrderedDict()
    sel = torch.nonzero(torch.isin(kk, heads_k)).reshape(-1)
    k_sel = kk[sel]
    perm = torch.sort(k_sel, stable=True).indices           # group by patch, factor-list order inside a group
    idx = sel[perm]
    xy = coords[0, idx, :, 0, 0].cpu()
    ks, counts = torch.unique_consecutive(k_sel[perm], return_counts=True)
    ks, counts = ks.cpu().tolist(), counts.cpu().tolist()
    start = dict(zip(ks, np.concatenate([[0], np.cumsum(counts)[:-1]]).tolist()))
    length = dict(zip(ks, counts))
    out = OrderedDict()
    for s_img, p_id in zip(ii[into].cpu().tolist(), heads_k.cpu().tolist()):
        if (s_img, p_id) in out:
            continue
        out[(s_img, p_id)] = xy[start[p_id]:start[p_id] + length[p_id]]
    return out


def _first_connected(ii, jj):
    """min target frame per source frame (reference :297 ``jj[ii==start_image].min()``)"""
    i_h, j_h = ii.cpu().numpy(), jj.cpu().numpy()
    first = {}
    order = np.argsort(i_h, kind="stable")
    i_s, j_s = i_h[order], j_h[order]
    cut = np.flatnonzero(np.diff(i_s)) + 1
    for seg_i, seg_j in zip(np.split(i_s, cut), np.split(j_s, cut)):
        first[int(seg_i[0])] = int(seg_j.min())
    return first


def fit_model_patch_track(next_frame_index, patch_dict, img_to_keyframe_map, ii, jj, data_shape, frequency=30, deg=2):
    """{track key: (spline x(t), spline y(t), weight of the predicted factor, last time stamp)} -- reference
    :289-322.  The last sample of every track (the reprojection into the virtual frame) is left out; the
    last PAST_PATCH_NUM samples are fitted with weights growing linearly in time."""
    height, width = data_shape
    first = _first_connected(ii, jj)
    stamps = (img_to_keyframe_map / frequency).cpu().numpy()      # torch true division: float32, as upstream
    models = OrderedDict()
    for key, track in patch_dict.items():
        start_image, _ = key
        x, y = track[:-1].T.cpu().numpy()
        t = stamps[first[start_image]:next_frame_index]
        inside = (x >= 0) & (x < width) & (y >= 0) & (y < height)
        masked_weights = 0 if np.all(inside[-PAST_PATCH_NUM:] == False) else 10 ** -9   # noqa: E712 (as upstream)
        x_, y_, t_ = x[-PAST_PATCH_NUM:], y[-PAST_PATCH_NUM:], t[-PAST_PATCH_NUM:]
        w = (t_ - t_[0]) / (t[-1] - t_[0]) + 10 ** -7
        assert len(t_) == len(x_)
        spl_x = UnivariateSpline(x=t_, y=x_, w=w, bbox=[None, None], k=deg, s=None, ext=0, check_finite=False)
        spl_y = UnivariateSpline(x=t_, y=y_, w=w, bbox=[None, None], k=deg, s=None, ext=0, check_finite=False)
        models[key] = (spl_x, spl_y, masked_weights, t_[-1])
    return models


def predict_patch_on_model(patch_models, step_to_pred_future, frequency, next_frame_index, coords, weights, ii, jj,
                           kk, reference_layout=True):
    """evaluate every track model ``step`` frames ahead and write a 3x3 grid around the prediction into the
    coords of the patch's factor into frame ``next_frame_index`` (in place, like the reference :325-345), set
    that factor's weight.  Returns (coords, weights).

    reference_layout=True writes what the upstream lines write: ``stack((rows_grid, cols_grid))`` of
    ``meshgrid(x, y)``, i.e. channel 0 = y + column offset, channel 1 = x + row offset -- transposed with
    respect to every other coords tensor of the code base (channel 0 = x).  False writes channel 0 = x."""
    if not patch_models:
        return coords, weights
    keys = list(patch_models.keys())
    new_xy = np.empty((len(keys), 2), np.float64)
    w_new = np.empty(len(keys), np.float64)
    for n, key in enumerate(keys):
        spl_x, spl_y, masked_weights, last_t = patch_models[key]
        new_time = last_t + (step_to_pred_future / frequency)
        new_xy[n, 0], new_xy[n, 1] = spl_x(new_time), spl_y(new_time)
        w_new[n] = masked_weights
    off = torch.arange(3, dtype=torch.float64)
    xs = (torch.from_numpy(new_xy[:, 0]) - 1)[:, None] + off          # arange(new - 1, new + 2)[:3]
    ys = (torch.from_numpy(new_xy[:, 1]) - 1)[:, None] + off
    cols_grid = xs[:, :, None].expand(-1, 3, 3)                        # meshgrid(x, y), 'ij': [i, j] -> x[i]
    rows_grid = ys[:, None, :].expand(-1, 3, 3)                        #                         [i, j] -> y[j]
    grid = torch.stack((rows_grid, cols_grid) if reference_layout else (cols_grid, rows_grid), dim=1)
    # the factor of each modelled patch into the virtual frame
    into = torch.nonzero(jj == next_frame_index).reshape(-1)
    k_into = kk[into].cpu().tolist()
    pos = {k: e for k, e in zip(k_into, into.cpu().tolist())}
    edge = torch.tensor([pos[p_id] for _, p_id in keys], dtype=torch.long, device=coords.device)
    coords[0, edge] = grid.to(coords.dtype).to(coords.device)
    weights[0, edge] = torch.from_numpy(w_new).to(weights.dtype).to(weights.device)[:, None]
    return coords, weights
