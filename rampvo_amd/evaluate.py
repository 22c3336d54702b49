"""Sequence-level harness around Ramp_vo (reference: evaluate.py:73-95, 185-260, 263-306; ramp/utils.py:633-656).

``run`` / ``run_pose_pred`` make the same calls in the same order as the reference's functions of the same
name, on any iterable of ``(image, events, intrinsics, mask)``.  The reference scores a run with evo
(``main_ape.ape(pose_relation=translation_part, align=True, correct_scale=True)['rmse']``, evaluate.py:295-304)
-- a third-party package that is not in this image; ``ate_rmse`` is that definition written out: Umeyama
Sim(3) alignment of the estimated positions onto the reference positions, then the RMS position error.
``Trajectory`` carries the three attributes of evo's PoseTrajectory3D the writers use.
"""
import os
import os.path as osp
from pathlib import Path

import numpy as np
import torch

from .Ramp_vo import Ramp_vo


class Trajectory:
    """positions_xyz [T,3], orientations_quat_wxyz [T,4], timestamps [T] (evo.core.trajectory.PoseTrajectory3D's
    fields as used by evaluate.py:86-95 and ramp/utils.py:641-645)"""

    def __init__(self, positions_xyz, orientations_quat_wxyz, timestamps):
        self.positions_xyz = np.asarray(positions_xyz, dtype=float)
        self.orientations_quat_wxyz = np.asarray(orientations_quat_wxyz, dtype=float)
        self.timestamps = np.asarray(timestamps, dtype=float)
        assert len(self.positions_xyz) == len(self.orientations_quat_wxyz) == len(self.timestamps)

    @classmethod
    def from_terminate(cls, poses, tstamps):
        """poses [T,7] = (tx ty tz qx qy qz qw) as returned by Ramp_vo.terminate() (evaluate.py:276-281 reorders
        the quaternion the same way)"""
        poses = np.asarray(poses, dtype=float)
        return cls(poses[:, :3], poses[:, [6, 3, 4, 5]], tstamps)

    @property
    def num_poses(self):
        return len(self.timestamps)


@torch.no_grad()
def run(cfg_VO, network, eval_cfg, data_list, ht=480, wd=640, device="cuda", inputs_ready="stream"):
    """reference evaluate.py:232-260 (without the dataset-specific resize): returns poses, tstamps, points, colors.
    The loop hands the tracker tensors that were produced on the current stream right before the call, as the
    reference's loop does; ``inputs_ready = "stream"`` lets the frames pipeline anyway (Ramp_vo.__init__: the tracker
    orders its front end behind the caller's stream with an event and runs on its own stream; results are identical).
    ``inputs_ready=False``: everything on the caller's stream, frame after frame."""
    train_cfg = eval_cfg["data_loader"]["train"]["args"]
    slam = Ramp_vo(cfg=cfg_VO, network=network, train_cfg=train_cfg, ht=ht, wd=wd, device=device)
    slam.inputs_ready = inputs_ready
    for t, (image, events, intrinsics, mask) in enumerate(data_list):
        slam(t, input_tensor=(events, image, mask), intrinsics=intrinsics)
    for _ in range(12):
        slam.update()
    points = slam.points_.cpu().numpy()[:slam.m]
    colors = slam.colors_.view(-1, 3).cpu().numpy()[:slam.m]
    poses, tstamps = slam.terminate()
    return poses, tstamps, points, colors


@torch.no_grad()
def run_pose_pred(cfg_VO, network, eval_cfg, data_list, t_horizon_to_pred, t_to_pred, deg_approx=4, ht=480, wd=640,
                  device="cuda", corrected=False):
    """reference evaluate.py:185-229: track up to frame t_to_pred, then extrapolate virtual keyframes for
    t_horizon_to_pred frames; returns terminate()'s (poses, tstamps).  corrected=False reproduces upstream's pose
    prediction bugs included (Ramp_vo.predict_future_pose's docstring); corrected=True opts out of them"""
    train_cfg = eval_cfg["data_loader"]["train"]["args"]
    slam = Ramp_vo(cfg=cfg_VO, network=network, train_cfg=train_cfg, ht=ht, wd=wd, device=device)
    last_keyframe_number = 0
    for t, (image, events, intrinsics, mask) in enumerate(data_list):
        if t < t_to_pred or t_to_pred < 0:
            slam(t, input_tensor=(events, image, mask), intrinsics=intrinsics)
            last_keyframe_number = slam.n
        if t == t_to_pred and t_to_pred > 0:
            for _ in range(12):
                slam.update()
        if t >= t_to_pred and t_to_pred > 0:
            slam.predict_future_pose(last_keyframe_number=last_keyframe_number, sec_to_pred_future=t - t_to_pred,
                                     abs_time=t, deg=deg_approx, corrected=corrected)
        if t == t_to_pred + t_horizon_to_pred:
            break
    for _ in range(12):
        slam.update()
    return slam.terminate()


# ------------------------------------------------------------------------ metric
def umeyama_sim3(src, dst):
    """least-squares s, R, t with dst ~ s R src + t (Umeyama 1991; evo's align(correct_scale=True))"""
    src, dst = np.asarray(src, float), np.asarray(dst, float)
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / len(src)
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    var_s = (xs ** 2).sum() / len(src)
    s = float(np.trace(np.diag(D) @ S) / var_s) if var_s > 0 else 1.0
    return s, R, mu_d - s * R @ mu_s


def ate_rmse(est_xyz, ref_xyz):
    """ATE as the reference reports it (evaluate.py:295-304): RMS translation error after Sim(3) alignment"""
    est_xyz, ref_xyz = np.asarray(est_xyz, float), np.asarray(ref_xyz, float)
    assert est_xyz.shape == ref_xyz.shape and est_xyz.shape[0] >= 3
    s, R, t = umeyama_sim3(est_xyz, ref_xyz)
    err = (s * (R @ est_xyz.T).T + t) - ref_xyz
    return float(np.sqrt((err ** 2).sum(1).mean()))


# ----------------------------------------------------------------------- writers
def save_results(traj_ref, traj_est, scene, j=0, eval_type="None", root=None):
    """stamped_groundtruth.txt / stamped_traj_estimate.txt: time[s] x y z qw qx qy qz (reference evaluate.py:73-95)"""
    save_dir = osp.join(root or os.getcwd(), "trajectory_evaluation", f"{eval_type}", "trial_" + str(j), scene)
    os.makedirs(save_dir, exist_ok=True)
    for name, tr in (("stamped_groundtruth.txt", traj_ref), ("stamped_traj_estimate.txt", traj_est)):
        time_s = (tr.timestamps * 10 ** -9)[..., np.newaxis]
        np.savetxt(osp.join(save_dir, name), np.concatenate((time_s, tr.positions_xyz, tr.orientations_quat_wxyz), axis=1))
    return save_dir


def save_output_for_COLMAP(name, traj, points, colors, fx, fy, cx, cy, H=480, W=640):
    """images.txt / points3D.txt / cameras.txt of a COLMAP text model (reference ramp/utils.py:633-656; x10 scale
    for visualisation, colours given in [0, 1])"""
    colmap_dir = Path(name)
    colmap_dir.mkdir(exist_ok=True, parents=True)
    scale = 10
    images = ""
    for idx, (x, y, z), (qw, qx, qy, qz) in zip(range(1, traj.num_poses + 1), traj.positions_xyz * scale,
                                                 traj.orientations_quat_wxyz):
        images += f"{idx} {qw} {qx} {qy} {qz} {x} {y} {z} 1\n\n"
    (colmap_dir / "images.txt").write_text(images)
    points3D = ""
    colors_uint = (np.asarray(colors) * 255).astype(np.uint8).tolist()
    for i, (p, c) in enumerate(zip((np.asarray(points) * scale).tolist(), colors_uint), start=1):
        points3D += f"{i} " + ' '.join(map(str, p + c)) + " 0.0 0 0 0 0 0 0\n"
    (colmap_dir / "points3D.txt").write_text(points3D)
    (colmap_dir / "cameras.txt").write_text(f"1 PINHOLE {W} {H} {fx} {fy} {cx} {cy}")
    return colmap_dir
