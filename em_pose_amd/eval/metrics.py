"""
Evaluation metrics (mirror of reference empose/eval/metrics.py:18-346): MPJPE, Procrustes-aligned MPJPE and the mean
global joint-angle error, accumulated per frame and reduced exactly like the reference (`get_metrics`,
metrics.py:289-330: mean over frames per joint, then mean over the evaluated joints; std over all selected entries).

This is host-side code in the reference (a NumPy SVD per frame in a Python loop, numpy-quaternion).  Here GPU tensors
take the device path: joints-only forward kinematics (`SMPLLayer.fk_joints`) and ONE kernel per call
that produces the per-frame Euclidean / Procrustes / global-angle rows (`empose_metrics_rows`, csrc/metrics.hip); CPU
tensors take the NumPy path below (same algorithm, used by the CPU tests).  The accumulator is a mergeable value: `state()` / `merge()` / `gather()` let every rank of a sequence-sharded evaluation keep its own
engine and combine them with ONE all_gather at the end -- the raw per-frame rows are exchanged so that
the reference's `np.std` is reproduced exactly, not approximated from moments.

The geodesic joint-angle distance is computed from rotation matrices, acos((tr(R1^T R2) - 1) / 2), which equals
numpy-quaternion's `rotation_intrinsic_distance` used by the reference (metrics.py:158); the device kernel builds the
global orientations with the reference's clamped exponential map (helpers/so3.py:86-128).
"""
import numpy as np
import torch

from em_pose_amd.helpers.configuration import CONSTANTS as C

EUCL_EVAL_JOINTS = ['root', 'l_hip', 'r_hip', 'spine1', 'l_knee', 'r_knee', 'spine2', 'l_ankle', 'r_ankle', 'spine3',
                    'neck', 'l_collar', 'r_collar', 'head', 'l_shoulder', 'r_shoulder', 'l_elbow', 'r_elbow',
                    'l_wrist', 'r_wrist']
ANGLE_EVAL_JOINTS = ['l_hip', 'r_hip', 'spine1', 'l_knee', 'r_knee', 'spine2', 'spine3', 'neck', 'l_collar',
                     'r_collar', 'head', 'l_shoulder', 'r_shoulder', 'l_elbow', 'r_elbow']


def procrustes_align(X, Y):
    """
    Similarity-align Y (J,3) onto X (J,3) (optimal rotation, scale, translation); returns the transformed Y.
    Same algorithm as the reference's `_procrustes` with compute_optimal_scale=True (metrics.py:18-66), batched.
    X, Y: (N,J,3) float64.
    """
    muX, muY = X.mean(1, keepdims=True), Y.mean(1, keepdims=True)
    X0, Y0 = X - muX, Y - muY
    normX = np.sqrt((X0 ** 2).sum((1, 2), keepdims=True))
    normY = np.sqrt((Y0 ** 2).sum((1, 2), keepdims=True))
    X0, Y0 = X0 / normX, Y0 / normY
    A = np.swapaxes(X0, 1, 2) @ Y0
    U, s, Vt = np.linalg.svd(A, full_matrices=False)
    V = np.swapaxes(Vt, 1, 2)
    T = V @ np.swapaxes(U, 1, 2)
    sign = np.sign(np.linalg.det(T))
    V[:, :, -1] *= sign[:, None]
    s[:, -1] *= sign
    T = V @ np.swapaxes(U, 1, 2)
    trace = s.sum(1)[:, None, None]
    return normX * trace * (Y0 @ T) + muX


def rotvec_to_matrix(r):
    """(...,3) -> (...,3,3), float64, exact small-angle handling."""
    r = np.asarray(r, dtype=np.float64)
    theta = np.linalg.norm(r, axis=-1)[..., None, None]
    K = np.zeros(r.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -r[..., 2], r[..., 1]
    K[..., 1, 0], K[..., 1, 2] = r[..., 2], -r[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -r[..., 1], r[..., 0]
    small = theta < 1e-8
    th = np.where(small, 1.0, theta)
    a = np.where(small, 1.0 - theta ** 2 / 6.0, np.sin(th) / th)
    b = np.where(small, 0.5 - theta ** 2 / 24.0, (1.0 - np.cos(th)) / th ** 2)
    return np.eye(3) + a * K + b * (K @ K)


def local_to_global_rotations(poses, parents):
    """Relative axis-angle joints (N, J*3) -> global rotation matrices (N,J,3,3) (reference utils.py:165-199)."""
    n_joints = poses.shape[-1] // 3
    local = rotvec_to_matrix(poses.reshape(-1, n_joints, 3))
    out = np.zeros_like(local)
    for j in range(n_joints):
        out[:, j] = local[:, j] if parents[j] < 0 else out[:, parents[j]] @ local[:, j]
    return out


def geodesic_degrees(Ra, Rb):
    tr = np.einsum('...ij,...ij->...', Ra, Rb)
    return np.rad2deg(np.arccos(np.clip((tr - 1.0) * 0.5, -1.0, 1.0)))


class MetricsEngine(object):
    def __init__(self, smpl_model):
        self.smpl_model = smpl_model
        self.eucl_idxs = [C.SMPL_JOINTS.index(j) for j in EUCL_EVAL_JOINTS]
        self.angle_idxs = [C.SMPL_JOINTS.index(j) - 1 for j in ANGLE_EVAL_JOINTS]
        self.angle_glob = True
        self.reset()

    def reset(self):
        self._eucl, self._eucl_pa, self._angle = [], [], []
        self._pending = []

    # the accumulators of the reference (lists of per-frame rows); reading them brings pending device rows to the host
    @property
    def eucl_dists(self):
        self._flush()
        return self._eucl

    @property
    def eucl_dists_pa(self):
        self._flush()
        return self._eucl_pa

    @property
    def angle_diffs(self):
        self._flush()
        return self._angle

    # ---- accumulation -------------------------------------------------------------------------------------------
    @staticmethod
    def valid_frames(seq_lengths, n, f, frame_mask=None):
        """The frames `compute` counts -- inside the sequence and with every sensor present -- from host tensors."""
        return MetricsEngine._mask(seq_lengths.cpu() if seq_lengths is not None else None, n, f,
                                   frame_mask.cpu() if frame_mask is not None else None, 'cpu')

    @staticmethod
    def _mask(seq_lengths, n, f, frame_mask, device):
        if seq_lengths is not None:
            mask = torch.arange(f, device=seq_lengths.device)[None, :] < seq_lengths.reshape(-1, 1)
        else:
            mask = torch.ones(n, f, dtype=torch.bool)
        mask = mask.to(dtype=torch.bool, device=device)
        if frame_mask is not None:
            fm = frame_mask.to(dtype=torch.bool, device=device)
            if fm.dim() == 3:
                fm = fm.logical_not().any(dim=-1).logical_not()
            mask = torch.logical_and(mask, fm)
        return mask

    def _add_device_rows(self, kp3d, kp3d_hat, pose=None, pose_hat=None, valid=None):
        """Euclidean / Procrustes / angle rows from the HIP kernel (empose_metrics_rows); `valid` (bool per row, on the
        device) selects the rows that count when they are brought to the host."""
        import ctypes
        from em_pose_amd import _lib
        n, dev = kp3d.shape[0], kp3d.device
        f32 = lambda t: None if t is None else t.to(dtype=torch.float32).contiguous()
        kp3d, kp3d_hat, pose, pose_hat = f32(kp3d), f32(kp3d_hat), f32(pose), f32(pose_hat)
        rows = torch.empty(n, 65, dtype=torch.float64, device=dev)
        parents = (ctypes.c_int * 22)(*C.SMPL_PARENTS)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().empose_metrics_rows(n, _lib.dptr(kp3d), _lib.dptr(kp3d_hat), _lib.dptr(pose),
                                                      _lib.dptr(pose_hat), parents, _lib.dptr(rows),
                                                      _lib.current_stream()))
        # kept on the device until somebody reads the accumulators: no host round trip per chunk
        self._pending.append((rows, pose is not None, valid))

    def take_device_rows(self):
        """The rows of the `compute` calls since the last read, still on the device and no longer this engine's:
        [(rows (m, 65) float64 = 22 Euclidean | 22 Procrustes | 21 angle columns, has_angle, valid), ...] in call order.
        For a driver that places and reads the rows itself (eval/helpers.py::evaluate_sequences_batched)."""
        pending, self._pending = self._pending, []
        return pending

    def _flush(self):
        """Device rows of earlier `compute` calls -> the host accumulators, in call order: ONE gather of the rows that
        count on the device (the per-call tensors concatenated, the valid rows picked by an index built on the host) and
        ONE copy into pinned host memory (torch caches pinned blocks: no page faults of a fresh multi-MB host array per
        call -- what made equal passes of the batched evaluation driver differ by 2x)."""
        pending, self._pending = self._pending, []
        if not pending:
            return
        if len({has_angle for _, has_angle, _ in pending}) > 1:   # mixed families: one by one
            for entry in pending:
                self._pending = [entry]
                self._flush()
            return
        has_angle = pending[0][1]
        dev = pending[0][0].device
        idx, offset, picked = [], 0, False
        for rows, _, valid in pending:
            m = rows.shape[0]
            if valid is None:
                idx.append(np.arange(offset, offset + m, dtype=np.int64))
            else:
                v = valid.cpu().numpy().reshape(-1)
                picked = picked or not v.all()
                idx.append(offset + np.flatnonzero(v))
            offset += m
        idx = np.concatenate(idx)
        if idx.shape[0] == 0:
            return
        rows = pending[0][0] if len(pending) == 1 else torch.cat([r for r, _, _ in pending])
        if picked:
            rows = rows.index_select(0, torch.from_numpy(idx).to(dev))
        if rows.is_cuda:
            host = torch.empty(rows.shape, dtype=rows.dtype, pin_memory=True)
            host.copy_(rows)
            rows = host.numpy()     # (the view keeps the pinned block alive; it returns to torch's cache with it)
        else:
            rows = rows.numpy()
        self._eucl.append(rows[:, :22])
        self._eucl_pa.append(rows[:, 22:44])
        if has_angle:
            self._angle.append(rows[:, 44:])

    def _add_eucl(self, kp3d, kp3d_hat):
        gt = kp3d.detach().cpu().numpy().astype(np.float64)
        hat = kp3d_hat.detach().cpu().numpy().astype(np.float64)
        self.eucl_dists.append(np.sqrt(((gt - hat) ** 2).sum(-1)))
        self.eucl_dists_pa.append(np.sqrt(((gt - procrustes_align(gt, hat)) ** 2).sum(-1)))

    def compute_joint_dist(self, joints, joints_hat, seq_lengths=None, frame_mask=None):
        n, f = joints.shape[0], joints.shape[1]
        mask = self._mask(seq_lengths, n, f, frame_mask, joints.device)
        if mask.sum() == 0:
            return
        js = joints[mask].reshape(-1, joints.shape[-1] // 3, 3)[:, :C.N_JOINTS + 1]
        js_hat = joints_hat[mask].reshape(-1, joints_hat.shape[-1] // 3, 3)[:, :C.N_JOINTS + 1]
        if js.is_cuda:
            self._add_device_rows(js, js_hat)
        else:
            self._add_eucl(js, js_hat)

    def compute(self, pose, shape, pose_hat, shape_hat=None, seq_lengths=None, pose_root=None, pose_root_hat=None,
                frame_mask=None, valid=None):
        """Same arguments as the reference (metrics.py:183-241).  `valid` (optional, bool (n, f), any device): the
        frames that count, for a caller that already has `seq_lengths` / `frame_mask` on the host (`valid_frames`) --
        the device path then launches nothing for the mask."""
        n, f = pose.shape[0], pose.shape[1]
        shape_hat = shape if shape_hat is None else shape_hat
        mask = valid if valid is not None else self._mask(seq_lengths, n, f, frame_mask, pose.device)
        if pose.is_cuda and self.angle_glob and hasattr(self.smpl_model, 'fk_joints'):
            # device path: joints-only forward kinematics + one metrics kernel over ALL n * f frames;
            # the valid rows are picked when the accumulators are read (`_flush`), so nothing here waits for the
            # device -- no compaction by a boolean mask, no emptiness test
            flat = lambda t: t.reshape(n * f, -1)
            per_frame = lambda s_: flat(s_) if s_.dim() == 3 else flat(s_.unsqueeze(1).expand(n, f, s_.shape[-1]))
            zeros = torch.zeros(n * f, 3, dtype=pose.dtype, device=pose.device)
            root_f = zeros if pose_root is None else flat(pose_root)
            root_hat_f = zeros if pose_root is None else flat(pose_root_hat)
            # ground truth and estimate through ONE forward-kinematics launch
            both = self.smpl_model.fk_joints(torch.cat([flat(pose), flat(pose_hat)]),
                                             torch.cat([per_frame(shape), per_frame(shape_hat)]),
                                             poses_root=torch.cat([root_f, root_hat_f]))
            self._add_device_rows(both[:n * f], both[n * f:], flat(pose), flat(pose_hat), valid=mask.reshape(n * f))
            return
        if valid is not None:
            mask = mask.to(pose.device)
        if mask.sum() == 0:
            return

        def shapes(s):
            return s[mask] if s.dim() == 3 else s.unsqueeze(1).repeat(1, f, 1)[mask]
        shape_f, shape_hat_f = shapes(shape), shapes(shape_hat)
        pose_f, pose_hat_f = pose[mask], pose_hat[mask]
        if pose_root is None:
            root_f = torch.zeros(pose_f.shape[0], 3, dtype=pose_f.dtype, device=pose_f.device)
            root_hat_f = torch.zeros_like(root_f)
        else:
            root_f, root_hat_f = pose_root[mask], pose_root_hat[mask]
        if self.smpl_model is not None:
            # no body model (the CPU plumbing configuration has no HIP device to evaluate SMPL-H on): angle metric only
            _, kp3d = self.smpl_model.fk(pose_f.contiguous(), shape_f.contiguous(), poses_root=root_f.contiguous(),
                                         window_size=1000)
            _, kp3d_hat = self.smpl_model.fk(pose_hat_f.contiguous(), shape_hat_f.contiguous(),
                                             poses_root=root_hat_f.contiguous(), window_size=1000)
            self._add_eucl(kp3d[:, :C.N_JOINTS + 1], kp3d_hat[:, :C.N_JOINTS + 1])
        p = pose_f.detach().cpu().numpy().astype(np.float64)
        ph = pose_hat_f.detach().cpu().numpy().astype(np.float64)
        if self.angle_glob:
            zeros = np.zeros((p.shape[0], 3))
            g = local_to_global_rotations(np.concatenate([zeros, p], -1), C.SMPL_PARENTS)[:, 1:]
            gh = local_to_global_rotations(np.concatenate([zeros, ph], -1), C.SMPL_PARENTS)[:, 1:]
        else:
            g = rotvec_to_matrix(p.reshape(p.shape[0], -1, 3))
            gh = rotvec_to_matrix(ph.reshape(ph.shape[0], -1, 3))
        self.angle_diffs.append(geodesic_degrees(g, gh))

    def compute_angle_dist(self, pose, pose_hat, seq_lengths=None, frame_mask=None, rep='aa'):
        """Joint-angle metric only, on the angles as given (no kinematic chain; reference metrics.py:267-287):
        pose / pose_hat (N, F, J*3) axis-angle or (N, F, J*9) rotation matrices."""
        if rep not in ('aa', 'rotmat'):
            raise ValueError("rep is 'aa' or 'rotmat'")
        n, f = pose.shape[0], pose.shape[1]
        mask = self._mask(seq_lengths, n, f, frame_mask, pose.device)
        if mask.sum() == 0:
            return
        p = pose[mask].detach().cpu().numpy().astype(np.float64)
        ph = pose_hat[mask].detach().cpu().numpy().astype(np.float64)
        if rep == 'aa':
            g, gh = rotvec_to_matrix(p.reshape(p.shape[0], -1, 3)), rotvec_to_matrix(ph.reshape(ph.shape[0], -1, 3))
        else:
            g, gh = p.reshape(p.shape[0], -1, 3, 3), ph.reshape(ph.shape[0], -1, 3, 3)
        self.angle_diffs.append(geodesic_degrees(g, gh))

    @staticmethod
    def to_tensorboard_log(metrics, writer, global_step, prefix=''):
        """reference metrics.py:342-346"""
        writer.add_scalar('metrics/{}/mje mean'.format(prefix), metrics['MPJPE [mm]'], global_step)
        writer.add_scalar('metrics/{}/mje pa mean'.format(prefix), metrics['PA-MPJPE [mm]'], global_step)
        writer.add_scalar('metrics/{}/mae mean'.format(prefix), metrics['MPJAE [deg]'], global_step)

    # ---- mergeable state ------------------------------------------------------------------------------------------
    def state(self):
        cat = lambda xs, w: np.concatenate(xs, axis=0) if xs else np.zeros((0, w))
        return {'eucl': cat(self.eucl_dists, C.N_JOINTS + 1), 'eucl_pa': cat(self.eucl_dists_pa, C.N_JOINTS + 1),
                'angle': cat(self.angle_diffs, C.N_JOINTS)}

    def merge(self, state):
        for key, store in (('eucl', self.eucl_dists), ('eucl_pa', self.eucl_dists_pa), ('angle', self.angle_diffs)):
            if state[key].shape[0]:
                store.append(np.asarray(state[key], dtype=np.float64))

    def gather(self, group=None, device=None, force=False):
        """
        Combine the accumulators of all ranks (torch.distributed; RCCL on GPUs, gloo on CPU). Rows are concatenated in
        rank order, so every rank ends up with the same, order-deterministic state.  `force` runs the collectives even
        in a group of one (self-test of the RCCL path on a single GPU).
        """
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
            return self
        world = dist.get_world_size(group)
        st = self.state()
        dev = torch.device('cpu') if device is None else device
        merged = {}
        for key in ('eucl', 'eucl_pa', 'angle'):
            mine = torch.from_numpy(st[key]).to(dev)
            count = torch.tensor([mine.shape[0]], dtype=torch.int64, device=dev)
            counts = [torch.zeros_like(count) for _ in range(world)]
            dist.all_gather(counts, count, group=group)
            counts = [int(c.item()) for c in counts]
            width = mine.shape[1]
            padded = torch.zeros(max(max(counts), 1), width, dtype=torch.float64, device=dev)
            padded[:mine.shape[0]] = mine
            parts = [torch.zeros_like(padded) for _ in range(world)]
            dist.all_gather(parts, padded, group=group)
            merged[key] = np.concatenate([p[:c].cpu().numpy() for p, c in zip(parts, counts)], axis=0)
        self.reset()
        self.merge(merged)
        return self

    # ---- reduction ------------------------------------------------------------------------------------------------
    def get_metrics(self, eucl_idxs_select=True, angle_idxs_select=True):
        # A metric family without a single accumulated row is NaN, not 0: "0 mm" reads as a perfect score (e.g. the
        # position metrics of an engine built without a body model: the CPU plumbing configuration).
        nan = float('nan')
        out = {'MPJPE [mm]': nan, 'MPJPE STD': nan, 'PA-MPJPE [mm]': nan, 'PA-MPJPE STD': nan, 'MPJAE [deg]': nan,
               'MPJAE STD': nan}
        if self.eucl_dists:
            e, ep = np.concatenate(self.eucl_dists, 0), np.concatenate(self.eucl_dists_pa, 0)
            idx = self.eucl_idxs if eucl_idxs_select else list(range(e.shape[1]))
            out['MPJPE [mm]'] = float(np.mean(np.mean(e, axis=0)[idx]) * 1000.0)
            out['MPJPE STD'] = float(np.std(e[:, idx]) * 1000.0)
            out['PA-MPJPE [mm]'] = float(np.mean(np.mean(ep, axis=0)[idx]) * 1000.0)
            out['PA-MPJPE STD'] = float(np.std(ep[:, idx]) * 1000.0)
        if self.angle_diffs:
            a = np.concatenate(self.angle_diffs, 0)
            idx = self.angle_idxs if angle_idxs_select else list(range(a.shape[1]))
            out['MPJAE [deg]'] = float(np.mean(np.mean(a, axis=0)[idx]))
            out['MPJAE STD'] = float(np.std(a[:, idx]))
        return out

    @staticmethod
    def to_pretty_string(metrics, model_name):
        from tabulate import tabulate
        return tabulate([[model_name] + list(metrics.values())], headers=['Model'] + list(metrics.keys()))
