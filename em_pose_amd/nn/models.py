"""
Model construction and forward() for the LGD path (mirror of reference empose/nn/models.py).

`create_model(config, smpl_model)`                       reference models.py:23-33
`IterativeErrorFeedback.forward(batch, window_size, is_new_sequence)`   reference models.py:485-632
`IterativeErrorFeedback.backward(batch, model_out, writer, global_step)` reference models.py:634-688 (loss values)
`FeedForwardResNet`                                      reference models.py:166-262 (CPU plumbing config only)

The module keeps the reference's parameter tree (same `state_dict` keys) but does none of the arithmetic itself:
forward() packs the parameters once into an opaque HIP model (include/empose_hip.h) and then calls
`empose_lgd_forward` -- LSTM / MLP GEMMs on the fp32 matrix cores, SMPL-H restricted to the sensor sub-mesh, and the
residual gradient by a hand-derived reverse pass instead of autograd (so `torch.set_grad_enabled` is left alone,
unlike reference models.py:487).  There is no CPU path: CPU tensors raise.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from em_pose_amd import _lib
from em_pose_amd.bodymodels import tables as TB
from em_pose_amd.helpers.configuration import CONSTANTS as CONST
from em_pose_amd.nn import layers as _layers
from em_pose_amd.nn.layers import MLP, FeedForwardResidualBlock, RNNLayer, fill_dense_desc, linear_hip, linear_train
from em_pose_amd.nn.loss import mask_from_seq_lengths, normal_mse, padded_loss, reconstruction_loss  # noqa: F401


def create_model(config, *args):
    m_type = config.m_type
    if m_type == 'resnet':
        return FeedForwardResNet(config, *args)
    elif m_type in ('ief', 'lgd'):
        return IterativeErrorFeedback(config, *args)
    elif m_type == 'rnn':
        return SimpleRNN(config, *args)
    raise ValueError("Model type '{}' unknown.".format(m_type))


# Counts registrations of parameters / buffers / submodules anywhere in the process: what invalidates the per-network
# tensor lists cached by IterativeErrorFeedback._own_parameters (a Parameter assigned to a module attribute goes through
# register_parameter, so a replaced weight is seen; in-place changes are seen through _version / data_ptr in the key).
_REGISTRATIONS = [0]


def _count_registration(*_args):
    _REGISTRATIONS[0] += 1


for _register in ('register_module_parameter_registration_hook', 'register_module_buffer_registration_hook',
                  'register_module_module_registration_hook'):
    getattr(torch.nn.modules.module, _register)(_count_registration)


def _cat_windows(parts):
    """Per-window results along the frame axis; a single window is returned as it is (a view, no device copy)."""
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)


class _SmplSensorsFn(torch.autograd.Function):
    """Differentiable `get_estimated_real_markers` for the training path: forward and vector-Jacobian product are the
    HIP sub-mesh kernels (empose_smpl_sensors_fwd_bwd / empose_smpl_sensors_vjp); no autograd graph through SMPL."""

    @staticmethod
    def forward(ctx, net, pose, shape, offset_r, offset_t, frames_per_window):
        pos, ori, joints = net.get_estimated_real_markers(pose.detach(), shape.detach(), offset_r, offset_t,
                                                          frames_per_window=frames_per_window)
        ctx.net, ctx.F = net, frames_per_window
        ctx.save_for_backward(pose.detach(), shape.detach(), offset_r, offset_t)
        return pos, ori, joints

    @staticmethod
    def backward(ctx, d_pos, d_ori, d_joints):
        pose, shape, offset_r, offset_t = ctx.saved_tensors
        net, dev, T = ctx.net, pose.device, pose.shape[0]
        f32 = lambda t: t.to(dtype=torch.float32).contiguous()
        pose, shape, d_pos, d_ori, d_joints = f32(pose), f32(shape), f32(d_pos), f32(d_ori), f32(d_joints)
        lib = _lib.lib()
        with torch.cuda.device(dev):
            handle = net._ensure_handle(dev)
            g_pose = torch.empty(T, 66, dtype=torch.float32, device=dev)
            g_shape = torch.empty(T, 10, dtype=torch.float32, device=dev)
            nbytes = lib.empose_smpl_vjp_workspace_bytes(handle, T)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.empose_smpl_sensors_vjp(handle, T, ctx.F, _lib.dptr(pose), 66, _lib.dptr(shape), 10,
                                                   _lib.dptr(offset_r), _lib.dptr(offset_t), _lib.dptr(d_pos),
                                                   _lib.dptr(d_ori), _lib.dptr(d_joints), _lib.dptr(g_pose),
                                                   _lib.dptr(g_shape), _lib.dptr(ws), nbytes, _lib.current_stream()))
        return None, g_pose, g_shape, None, None, None


class _InjectGrad(torch.autograd.Function):
    """Identity whose backward adds a constant cotangent.  The reference calls `E_i.backward(retain_graph=True)` inside
    forward (models.py:576), which deposits dE_i/dparams through the producers of the current estimate.  Backward is
    linear, so adding dE_i/dpose_i (= the gradient feature / (B*F)) to whatever flows into pose_i during the single
    final backward gives the same parameter gradients without N extra backward passes."""

    @staticmethod
    def forward(ctx, x, g):
        ctx.save_for_backward(g)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        (g,) = ctx.saved_tensors
        return dy + g, None


# The per-sensor input features a configuration can switch on, in the order they are laid out in a network input row:
# (configuration flag, key of the input dict, floats per sensor).  Every size of the models follows from this table.
SENSOR_FEATURES = (('use_marker_pos', 'marker_pos', 3),
                   ('use_marker_ori', 'marker_oris', 9),
                   ('use_marker_nor', 'marker_normals', 3))


def pack_sensor_inputs(marker_pos, marker_oris, marker_idxs, marker_masks=None, seq_lengths=None, out=None,
                       want_frame_weight=False):
    """
    Device tensors (B,F,36) / (B,F,108) -> network input rows (B,F,12*len(marker_idxs)): the chosen sensors'
    positions, then their row-major orientations -- one launch of `empose_pack_inputs` (no index_select / cat chain).
    `out` may be a wider (B*F, ld) buffer whose leading columns are filled.  With `want_frame_weight` the per-frame
    weight of the in-loop residual (reference loss.py:31-39, models.py:578-579) comes out of the same launch.
    """
    dev, (B, F) = marker_pos.device, marker_pos.shape[:2]
    f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
    marker_pos, marker_oris = f32(marker_pos), f32(marker_oris)
    if marker_pos.numel() != B * F * 36 or marker_oris.numel() != B * F * 108:
        raise ValueError('expected 12 sensors per frame: marker_pos (B,F,36), marker_oris (B,F,108)')
    width = 12 * len(marker_idxs)
    x = torch.empty(B, F, width, dtype=torch.float32, device=dev) if out is None else out
    ldx = x.stride(-2)
    assert x.is_cuda and x.dtype == torch.float32 and x.stride(-1) == 1 and ldx >= width
    weight = torch.empty(B * F, dtype=torch.float32, device=dev) if want_frame_weight else None
    masks = None if marker_masks is None else f32(marker_masks)
    lens = None if seq_lengths is None else seq_lengths.to(device=dev, dtype=torch.int32).contiguous()
    idx = (C.c_int * 12)(*marker_idxs)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().empose_pack_inputs(B, F, len(marker_idxs), idx, _lib.dptr(marker_pos),
                                                 _lib.dptr(marker_oris), _lib.dptr(masks), _lib.dptr(lens),
                                                 _lib.dptr(x), ldx, _lib.dptr(weight), _lib.current_stream()))
    return (x, weight) if want_frame_weight else x


class BaseModel(nn.Module):
    """What the three model families share (contract of reference models.py:36-163: attribute names, `prepare_inputs`,
    `window_generator`, the `model_name` suffix).  Sizes come from SENSOR_FEATURES; on GPU tensors `prepare_inputs` is
    one pack launch."""

    def __init__(self, config, smpl_model=None):
        super(BaseModel, self).__init__()
        self.config, self.smpl = config, smpl_model
        configured = getattr(config, 'n_markers', -1)
        self.n_markers = configured if configured > -1 else CONST.N_TRACKERS_WO_ROOT
        self.n_frames = config.window_size
        self.estimate_shape, self.shape_avg = config.m_estimate_shape, config.m_average_shape
        self.fk_loss_weight = config.m_fk_loss
        self.do_fk = self.fk_loss_weight > 0.0
        self.pose_weight = getattr(config, 'm_pose_loss_weight', 1.0)
        self.shape_weight = getattr(config, 'm_shape_loss_weight', 1.0)
        if self.do_fk and (self.smpl is None or not (self.estimate_shape or isinstance(self, IterativeErrorFeedback))):
            raise AssertionError('an FK loss needs a body model and an estimated shape')
        self.set_input_output_size()
        self.create_model()

    def active_features(self):
        """Rows of SENSOR_FEATURES the configuration enables; orientations and normals exclude each other."""
        on = [row for row in SENSOR_FEATURES if getattr(self.config, row[0])]
        assert not (self.config.use_marker_ori and self.config.use_marker_nor)
        return on

    def set_input_output_size(self):
        self.input_size = self.n_markers * sum(width for _, _, width in self.active_features())
        self.output_size = (CONST.N_JOINTS + 1) * 3
        self.config.input_size, self.config.output_size = self.input_size, self.output_size

    def create_model(self):
        raise NotImplementedError('Must be implemented by subclass.')

    def sensor_subset(self):
        """Indices (into the 12 sensors of a batch) of the sensors this model reads."""
        if self.n_markers not in (6, 12):
            raise AssertionError('6 or 12 sensors')
        return list(range(12)) if self.n_markers == 12 else list(CONST.S_CONFIG_6)

    def prepare_inputs(self, batch_inputs):
        """Input dict of a batch -> (N, F, input_size) rows: per enabled feature, the model's sensors flattened."""
        feats = self.active_features()
        if any(flag == 'use_marker_nor' for flag, _, _ in feats):
            raise ValueError('Normals currently not supported.')
        pos, ori = batch_inputs['marker_pos'], batch_inputs['marker_oris']
        subset = self.sensor_subset()
        if pos.is_cuda and [flag for flag, _, _ in feats] == ['use_marker_pos', 'use_marker_ori']:
            return pack_sensor_inputs(pos, ori, subset)
        # CPU tensors (the ResNet plumbing configuration) or a single feature: plain indexing
        n, f = pos.shape[0], pos.shape[1]
        pick = torch.as_tensor(subset, dtype=torch.long, device=pos.device)
        cols = [batch_inputs[key].reshape(n, f, 12, width).index_select(2, pick).reshape(n, f, -1)
                for _, key, width in feats]
        return cols[0] if len(cols) == 1 else torch.cat(cols, dim=-1)

    def window_generator(self, batch, window_size, **input_options):
        """The batch as one piece (window_size None: every caller in the tree), or cut along time into pieces of
        `window_size` frames with a fresh one-entry `seq_lengths` -- valid for batch size 1 only, like the reference's
        (models.py:146-163)."""
        if window_size is None:
            pieces = [(None, None, batch.seq_lengths)]
        else:
            total = batch.seq_length
            one = lambda n: torch.tensor([n], dtype=batch.seq_lengths.dtype, device=batch.seq_lengths.device)
            pieces = [(sf, min(sf + window_size, total), one(min(sf + window_size, total) - sf))
                      for sf in range(0, total, window_size)]
        for sf, ef, lengths in pieces:
            batch_inputs = batch.get_inputs(**input_options) if sf is None else \
                batch.get_inputs(sf=sf, ef=ef, **input_options)
            batch_inputs['seq_lengths'] = lengths
            yield batch_inputs

    def model_name(self):
        """`-shape<h>[-avg][-fk<w>]-n<sensors>-lr<lr>`: the suffix of the baselines' names."""
        parts = []
        if self.estimate_shape is not None:
            parts.append('shape{}{}'.format(self.config.m_shape_hidden_size, '-avg' if self.shape_avg else ''))
        if self.do_fk:
            parts.append('fk{}'.format(self.fk_loss_weight))
        parts += ['n{}'.format(self.n_markers), 'lr{}'.format(self.config.lr)]
        return ''.join('-' + part for part in parts)

    def maybe_do_fk(self, pose_hat, shape_hat):
        """Joints of the predicted pose if an FK loss is configured (reference models.py:134-144)."""
        if not self.do_fk:
            return None
        n, f = pose_hat.shape[0], pose_hat.shape[1]
        joints = self.smpl.fk_joints(pose_hat[:, :, 3:].reshape(n * f, -1), shape_hat.reshape(n * f, -1),
                                     poses_root=pose_hat[:, :, :3].reshape(n * f, -1))
        return joints.reshape(n, f, -1)

    def _fk_train(self, pose_hat, shape_hat):
        """`maybe_do_fk` WITH a reverse pass, for the baselines in training mode: the 22 joints and their vector-Jacobian
        product are the sub-mesh kernels of the LGD path (`_SmplSensorsFn`: empose_smpl_sensors_fwd_bwd / _vjp) through a
        helper model handle that is built once; its sensors are not used (their cotangents are zero)."""
        if not self.do_fk:
            return None
        n, f = pose_hat.shape[0], pose_hat.shape[1]
        helper = getattr(self, '_fk_helper', None)
        if helper is None:
            from em_pose_amd.helpers.configuration import lgd_config
            # (kept out of the module tree: its parameters are not the baseline's)
            net = create_model(lgd_config(12, False, 1, hidden=32), self.smpl)
            ids = getattr(self, 'fk_vertex_ids', None)    # (a body model smaller than SMPL-H needs its own sensor sites)
            if ids is not None:
                net.vertex_ids = [int(v) for v in ids]
            helper = [net.to(pose_hat.device).eval()]
            object.__setattr__(self, '_fk_helper', helper)
        dev = pose_hat.device
        o_r = torch.eye(3, device=dev).expand(1, 12, 3, 3).contiguous()
        o_t = torch.zeros(1, 12, 3, device=dev)
        _, _, joints = _SmplSensorsFn.apply(helper[0], pose_hat.reshape(n * f, -1), shape_hat.reshape(n * f, -1), o_r, o_t,
                                            n * f)
        return joints.reshape(n, f, -1)

    def _heads(self, features):
        """pose head, optional shape MLP (+ per-window mean), optional FK: shared by the two baselines."""
        n, f = features.shape[0], features.shape[1]
        flat = features.reshape(n * f, -1).contiguous().float()
        train = self.training and flat.is_cuda
        if train:       # autograd Functions over the HIP GEMMs (nn/layers.py): reference models.py:202-216, 303-316
            pose_hat = linear_train(flat, self.to_pose).reshape(n, f, -1)
        elif flat.is_cuda:
            pose_hat = linear_hip(flat, self.to_pose).reshape(n, f, -1)
        else:
            pose_hat = self.to_pose(features)
        shape_hat = None
        if self.to_shape is not None:
            shape_hat = (self.to_shape(flat) if flat.is_cuda and not train else self.to_shape.forward_torch(flat)) \
                .reshape(n, f, -1)
            if self.shape_avg:
                shape_hat = torch.mean(shape_hat, dim=1, keepdim=True).repeat((1, f, 1))
        joints_hat = self._fk_train(pose_hat, shape_hat) if train else self.maybe_do_fk(pose_hat, shape_hat)
        return {'pose_hat': pose_hat[:, :, 3:], 'root_ori_hat': pose_hat[:, :, :3], 'shape_hat': shape_hat,
                'joints_hat': joints_hat}

    def _baseline_backward(self, batch, model_out, writer=None, global_step=None):
        """Loss values of the two baselines and, in training mode, `total_loss.backward()` (reference models.py:223-262,
        326-366): the outputs of a training-mode forward carry an autograd graph over the HIP kernels (linear layers,
        LSTM with back-propagation through time, joints with their vector-Jacobian product)."""
        pose_hat, root_hat, shape_hat = model_out['pose_hat'], model_out['root_ori_hat'], model_out['shape_hat']
        dev, n, f = pose_hat.device, batch.batch_size, batch.seq_length
        sl = batch.seq_lengths.to(dev)
        masks = batch.marker_masks.to(dev) if batch.marker_masks is not None else None
        pose_loss = normal_mse(batch.poses_body.to(dev).reshape(n, f, -1, 3), pose_hat.reshape(n, f, -1, 3), sl, masks)
        root_loss = normal_mse(batch.poses_root.to(dev).reshape(n, f, -1, 3), root_hat.reshape(n, f, -1, 3), sl, masks)
        shape_loss, fk_loss = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        if self.estimate_shape:
            shape_loss = padded_loss(batch.shapes.to(dev).unsqueeze(1).repeat((1, shape_hat.shape[1], 1)), shape_hat,
                                     self.shape_loss, sl)
        if self.do_fk:
            fk_loss = reconstruction_loss(batch.joints_gt.to(dev).reshape(n, f, -1, 3),
                                          model_out['joints_hat'].reshape(n, f, -1, 3), sl, masks)
        total = pose_loss + root_loss + shape_loss + self.fk_loss_weight * fk_loss
        loss_vals = {'pose': pose_loss.item(), 'root_pose': root_loss.item(), 'shape': shape_loss.item(),
                     'fk': fk_loss.item(), 'total_loss': total.item()}
        if writer is not None:
            self.log_loss_vals(loss_vals, writer, global_step)
        if self.training:
            if not total.requires_grad:
                raise RuntimeError('training mode, but the outputs carry no graph: they come from an eval-mode forward')
            total.backward()
        return total, loss_vals

    def log_loss_vals(self, loss_vals, writer, global_step):
        mode_prefix = 'train' if self.training else 'valid'
        for k in loss_vals:
            writer.add_scalar('{}/{}'.format(k, mode_prefix), loss_vals[k], global_step)


class FeedForwardResNet(BaseModel):
    """
    Frame-wise residual MLP baseline (reference models.py:166-262). On GPU tensors every layer is one fp32 MFMA
    launch (residual + ReLU in the epilogue); on CPU tensors it is plain PyTorch, because BASELINE.json configs[0] is
    "forward on PyTorch CPU ... plumbing, no GPU".
    """

    def create_model(self):
        cfg = self.config
        self.hidden_size, self.num_layers = cfg.m_hidden_size, cfg.m_num_layers
        self.from_input = nn.Linear(self.input_size, self.hidden_size)
        self.blocks = nn.Sequential(*[FeedForwardResidualBlock(self.hidden_size, self.hidden_size)
                                      for _ in range(self.num_layers)])
        self.to_pose = nn.Linear(self.hidden_size, self.output_size)
        self.to_shape = MLP(self.hidden_size, CONST.N_SHAPE_PARAMS, cfg.m_shape_hidden_size, 2, cfg.m_dropout_hidden,
                            cfg.m_skip_connections, use_batch_norm=False) if self.estimate_shape else None
        self.shape_loss = nn.L1Loss(reduction='none')

    def model_name(self):
        return 'ResNet-{}x{}'.format(self.num_layers, self.hidden_size) + super(FeedForwardResNet, self).model_name()

    def forward(self, batch, window_size=None, is_new_sequence=True):
        inputs_ = self.prepare_inputs(batch.get_inputs())
        if inputs_.is_cuda:
            n, f = inputs_.shape[0], inputs_.shape[1]
            flat = inputs_.reshape(n * f, -1).contiguous().float()
            x = linear_train(flat, self.from_input) if self.training else linear_hip(flat, self.from_input)
            x = self.blocks(x).reshape(n, f, -1)     # (training mode: the blocks build the graph themselves)
        else:
            x = self.blocks(self.from_input(inputs_))
        return self._heads(x)

    def backward(self, batch, model_out, writer=None, global_step=None):
        return self._baseline_backward(batch, model_out, writer, global_step)


class SimpleRNN(BaseModel):
    """The (Bi)RNN baseline (reference models.py:265-366): (Bi)LSTM -> linear pose head (+ shape MLP, + FK joints)."""

    def create_model(self):
        cfg = self.config
        hidden, dirs = cfg.m_hidden_size, (2 if cfg.m_bidirectional else 1)
        self.rnn = RNNLayer(self.input_size, hidden, cfg.m_num_layers, bidirectional=cfg.m_bidirectional,
                            dropout=cfg.m_dropout, learn_init_state=cfg.m_learn_init_state)
        self.to_pose = nn.Linear(hidden * dirs, self.output_size)
        self.to_shape = MLP(hidden * dirs, CONST.N_SHAPE_PARAMS, cfg.m_shape_hidden_size, 2, cfg.m_dropout_hidden,
                            cfg.m_skip_connections, use_batch_norm=False) if self.estimate_shape else None
        self.shape_loss = nn.L1Loss(reduction='none')

    def model_name(self):
        name = 'RNN-{}'.format('-'.join([str(self.config.m_hidden_size)] * self.config.m_num_layers))
        if self.config.m_bidirectional:
            name = 'Bi' + name
        return name + super(SimpleRNN, self).model_name()

    def forward(self, batch, window_size=None, is_new_sequence=True):
        if is_new_sequence:
            self.rnn.final_state = None
        self.rnn.init_state = self.rnn.final_state
        inputs_ = self.prepare_inputs(batch.get_inputs())
        if self.training and inputs_.is_cuda:
            lstm_out = self.rnn.forward_torch(inputs_.contiguous().float(), batch.seq_lengths)
        else:
            lstm_out = self.rnn(inputs_, batch.seq_lengths)
        return self._heads(lstm_out)

    def backward(self, batch, model_out, writer=None, global_step=None):
        return self._baseline_backward(batch, model_out, writer, global_step)


class IterativeErrorFeedback(BaseModel):
    """The LGD / LGD-RNN model."""

    def __init__(self, config, smpl_model):
        self.N = config.m_num_iterations
        self.step_size = config.m_step_size
        self.shape_avg = config.m_average_shape
        self.r_weight = config.m_reprojection_loss_weight
        self.use_gradient = config.m_use_gradient
        self.skip_connections = config.m_skip_connections
        self.rnn_init = config.m_rnn_init
        super(IterativeErrorFeedback, self).__init__(config, smpl_model)
        self.vertex_ids = list(CONST.VERTEX_IDS)
        self.helper_ids = None  # optional explicit helper-vertex table (data of a trained model), else derived
        self.marker_idxs = self.sensor_subset()
        self.keep_history = True
        # False: per-window shape mean over all F frames incl. padding, as the reference; True: over valid frames only
        self.shape_avg_valid_only = False
        self.keep_gradient_trace = False
        self.gradient_trace = None
        self.markers_hat_history = None
        self.markers_ori_hat_history = None
        self.pose_hat_history = None
        self.shape_hat_history = None
        self.joints_hat_history = None
        self._handle = None      # opaque empose_model_t*
        self._handle_key = None  # what it was built from
        self._smpl_handle = None      # body-model-only handle used by the training path
        self._smpl_handle_key = None
        self._workspace = None
        # Streaming (eval/helpers.py::evaluate_sequences): a side stream for the refinement iterations.  When set, a
        # forward runs input packing + initial estimate (the LSTM with its state carry) on the current stream and the N
        # iterations on this one, so that the NEXT chunk's LSTM -- which needs only this chunk's final LSTM state --
        # runs beside this chunk's iterations.  The outputs are then complete only when `self.outputs_ready` (an event
        # on the side stream, renewed by every forward) has passed: the caller waits for it before it reads them.
        self.iter_stream = None
        self.outputs_ready = None
        self._pipe = None   # two workspaces in turn + the event after which each may be reused

    def set_input_output_size(self):
        if self.config.use_marker_nor:
            raise ValueError('Normals currently not supported.')
        if not (self.config.use_marker_pos and self.config.use_marker_ori):
            raise NotImplementedError('the HIP path implements the released configuration: positions + orientations')
        self.pos_d_start, self.pos_d_end = 0, self.n_markers * 3
        self.ori_d_start, self.ori_d_end = self.pos_d_end, self.pos_d_end + self.n_markers * 9
        self.input_size = self.n_markers * 12
        self.pose_size = (CONST.N_JOINTS + 1) * 3
        self.shape_size = CONST.N_SHAPE_PARAMS
        self.input_iter_size = self.input_size + self.pose_size + self.shape_size
        if self.use_gradient:
            self.input_iter_size += self.pose_size + self.shape_size
        for k in ('input_size', 'pose_size', 'shape_size', 'input_iter_size'):
            setattr(self.config, k, getattr(self, k))

    def create_model(self):
        cfg = self.config
        if self.rnn_init:
            if cfg.m_rnn_bidirectional:
                raise NotImplementedError('bidirectional init RNN (the reference itself mis-sizes the heads for it)')
            self.rnn = RNNLayer(self.input_size, cfg.m_rnn_hidden_size, cfg.m_rnn_num_layers, dropout=cfg.m_dropout)
            self.pose_net_init = nn.Linear(cfg.m_rnn_hidden_size, self.pose_size)
            self.shape_net_init = nn.Linear(cfg.m_rnn_hidden_size, self.shape_size)
        else:
            mk = lambda out: MLP(self.input_size, out, cfg.m_hidden_size, cfg.m_num_layers, cfg.m_dropout_hidden,
                                 self.skip_connections, not cfg.m_no_batch_norm)
            self.pose_net_init = mk(self.pose_size)
            self.shape_net_init = mk(self.shape_size)
        mk = lambda out: MLP(self.input_iter_size, out, cfg.m_hidden_size, cfg.m_num_layers, cfg.m_dropout_hidden,
                             self.skip_connections, not cfg.m_no_batch_norm)
        self.pose_net_iter = mk(self.pose_size)
        self.shape_net_iter = mk(self.shape_size)
        self.smpl_loss = nn.L1Loss(reduction='none')

    def model_name(self):
        cfg = self.config
        name = 'IEF-{}x{}-N{}'.format(cfg.m_num_layers, cfg.m_hidden_size, cfg.m_num_iterations)
        if self.rnn_init:
            name += '-{}RNN-{}x{}'.format('Bi' if cfg.m_rnn_bidirectional else '', cfg.m_rnn_num_layers,
                                          cfg.m_rnn_hidden_size)
        name += '-r{}-ws{}-lr{}'.format(self.r_weight, cfg.window_size, cfg.lr)
        name += '-grad' if self.use_gradient else ''
        name += '-skip' if self.skip_connections else ''
        name += '-n{}'.format(self.n_markers)
        return name

    # ---- HIP model handle ------------------------------------------------------------------------------------
    def _own_parameters(self):
        # `named_parameters()` over the module tree costs ~0.5 ms per call (prefix strings, memo sets): more than the
        # host side of a whole streaming chunk.  The list is cached and thrown away whenever ANY module registers a
        # parameter, a buffer or a submodule (global registration hooks below: such events are construction-time only),
        # or tensors are replaced wholesale (_apply / load_state_dict).
        cached = self.__dict__.get('_own_cache')
        if cached is None or cached[0] != _REGISTRATIONS[0]:
            tensors = [p for n, p in self.named_parameters() if not n.startswith('smpl.')] + \
                      [b for n, b in self.named_buffers() if not n.startswith('smpl.')]
            cached = (_REGISTRATIONS[0], tensors)
            self.__dict__['_own_cache'] = cached
        return cached[1]

    def _apply(self, fn, *args, **kwargs):
        self.__dict__['_own_cache'] = None
        return super(IterativeErrorFeedback, self)._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.__dict__['_own_cache'] = None
        return super(IterativeErrorFeedback, self).load_state_dict(*args, **kwargs)

    def _rodrigues(self):
        return getattr(self.smpl, 'rodrigues_convention', 'smplx')

    def _state_key(self, device):
        ps = self._own_parameters()
        return (device.index, tuple(self.vertex_ids), tuple(self.helper_ids or ()), self.N, self.shape_avg_valid_only,
                (self._rodrigues(), getattr(self.smpl, 'tables_version', 0)), _layers.BN_STATS_GENERATION[0],
                tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps))

    def release(self):
        if self._handle is not None:
            _lib.lib().empose_model_destroy(self._handle)
            self._handle, self._handle_key = None, None
        for _, handle in getattr(self, '_alt_handles', {}).values():
            _lib.lib().empose_model_destroy(handle)
        self._alt_handles = {}

    def release_all(self):
        self.release()
        if getattr(self, '_smpl_handle', None) is not None:
            _lib.lib().empose_model_destroy(self._smpl_handle)
            self._smpl_handle, self._smpl_handle_key = None, None

    def __del__(self):
        try:
            self.release_all()
        except Exception:
            pass

    def sub_mesh_tables(self):
        return TB.build_lgd_tables(self.smpl.model, self.vertex_ids, self.helper_ids, CONST.N_SHAPE_PARAMS)

    def _ensure_handle(self, device):
        if self.training:
            return self._ensure_smpl_handle(device)
        key = self._state_key(device)
        if self._handle is not None and key == self._handle_key:
            return self._handle
        # A driver may alternate between the two shape-averaging modes (eval/helpers.py::evaluate_sequences_batched):
        # the handle of the other mode is parked instead of being destroyed and rebuilt on every switch.
        alt = getattr(self, '_alt_handles', None)
        if alt is None:
            alt = self._alt_handles = {}
        if self._handle is not None:
            old = alt.pop(self._handle_key[4], None)
            if old is not None:
                _lib.lib().empose_model_destroy(old[1])
            alt[self._handle_key[4]] = (self._handle_key, self._handle)
            self._handle, self._handle_key = None, None
        parked = alt.pop(self.shape_avg_valid_only, None)
        if parked is not None:
            if parked[0] == key:
                self._handle_key, self._handle = parked
                return self._handle
            _lib.lib().empose_model_destroy(parked[1])
        return self._build_handle(device, key, smpl_only=False)

    def _ensure_smpl_handle(self, device):
        """Body-model-only handle for the training path: its key ignores the network parameters, which change with
        every optimiser step."""
        key = (device.index, tuple(self.vertex_ids), tuple(self.helper_ids or ()), self._rodrigues(),
               getattr(self.smpl, 'tables_version', 0))
        if self._smpl_handle is not None and key == self._smpl_handle_key:
            return self._smpl_handle
        if self._smpl_handle is not None:
            _lib.lib().empose_model_destroy(self._smpl_handle)
        self._smpl_handle, self._smpl_handle_key = None, None
        return self._build_handle(device, key, smpl_only=True)

    def _build_handle(self, device, key, smpl_only):
        keep = []
        desc = _lib.ModelDesc()
        tab = self.sub_mesh_tables()
        s = desc.smpl
        for k in ('n_sensors', 'nv', 'j_off', 'ncp', 'kb', 'max_deg'):
            setattr(s, k, int(tab[k]))
        for k in ('wc', 'wct', 'skin_w', 'bone_w'):
            setattr(s, k, _lib.fptr(tab[k]))
        for k in ('parents', 'skin_idx', 'bone_ptr', 'bone_vert', 's_center', 's_helper', 's_deg', 's_faces',
                  'path_ptr', 'path', 'sub_ptr', 'sub'):
            setattr(s, k, _lib.iptr(tab[k]))
        s.rodrigues = _lib.RODRIGUES[self._rodrigues()]
        keep.append(tab)
        desc.n_markers = self.n_markers
        for i, v in enumerate(self.marker_idxs):
            desc.marker_idx[i] = v
        desc.step_size = float(self.step_size)
        desc.shape_avg = (2 if self.shape_avg_valid_only else 1) if self.shape_avg else 0
        desc.use_gradient = int(bool(self.use_gradient))
        if smpl_only:
            desc.n_iterations, desc.rnn_init = 0, 0
        else:
            desc.n_iterations = self.N
            desc.rnn_init = int(bool(self.rnn_init))
            if self.rnn_init:
                self.rnn.fill_desc(desc.rnn, keep)
                fill_dense_desc(desc.pose_head, self.pose_net_init, None, None, keep)
                fill_dense_desc(desc.shape_head, self.shape_net_init, None, None, keep)
            else:
                self.pose_net_init.fill_desc(desc.pose_init, keep)
                self.shape_net_init.fill_desc(desc.shape_init, keep)
            self.pose_net_iter.fill_desc(desc.pose_iter, keep)
            self.shape_net_iter.fill_desc(desc.shape_iter, keep)
        handle = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().empose_model_create(C.byref(desc), C.byref(handle)))
        if smpl_only:
            self._smpl_handle, self._smpl_handle_key = handle, key
        else:
            self._handle, self._handle_key = handle, key
        return handle

    def get_estimated_real_markers(self, poses, shapes, offset_r, offset_t, vertex_ids=None, frames_per_window=1):
        """
        Virtual sensors (with offsets applied) and body joints for given pose/shape (reference models.py:471-483).
        poses (T,66), shapes (T,10); offsets per window: offset_r (T/F,12,3,3), offset_t (T/F,12,3).
        :return: pos (T,12,3), ori (T,12,3,3), joints (T,22,3)
        """
        if vertex_ids is not None and list(vertex_ids) != list(self.vertex_ids):
            raise ValueError('the sensor sub-mesh is baked for self.vertex_ids; set it before the first call')
        if not poses.is_cuda:
            raise _lib.EmposeError('needs GPU tensors; there is no CPU fallback')
        dev, T, F = poses.device, poses.shape[0], int(frames_per_window)
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        poses, shapes, offset_r, offset_t = f32(poses), f32(shapes), f32(offset_r), f32(offset_t)
        assert T % F == 0 and offset_r.shape[0] == T // F and offset_t.shape[0] == T // F
        lib = _lib.lib()
        with torch.cuda.device(dev):
            handle = self._ensure_handle(dev)
            pos = torch.empty(T, 12, 3, dtype=torch.float32, device=dev)
            ori = torch.empty(T, 12, 3, 3, dtype=torch.float32, device=dev)
            joints = torch.empty(T, 22, 3, dtype=torch.float32, device=dev)
            nbytes = lib.empose_smpl_workspace_bytes(handle, T)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.empose_smpl_sensors_fwd_bwd(handle, T, F, _lib.dptr(poses), 66, _lib.dptr(shapes), 10,
                                                       _lib.dptr(offset_r), _lib.dptr(offset_t), None, 0, None,
                                                       _lib.dptr(pos), _lib.dptr(ori), _lib.dptr(joints), None, 0,
                                                       None, 0, _lib.dptr(ws), nbytes, _lib.current_stream()))
        return pos, ori, joints

    # ---- forward ---------------------------------------------------------------------------------------------
    def forward_tensors(self, marker_pos, marker_oris, offset_t, offset_r, marker_masks=None, seq_lengths=None,
                        state=None, keep_history=False, keep_gradient_trace=False, suppress_mask_value=None):
        """
        One window batch through `empose_lgd_forward`. All tensors on the GPU, fp32.
        `suppress_mask_value` (a float): the readings are raw and the packing kernel replaces those of sensors whose
        mask is not 1 by that value (what `RealBatch.get_inputs` otherwise does before the model sees them).
        :return: dict(pose (B,F,66), shape (B,F,10), joints (B,F,66), state (h_n,c_n) or None, hist {...} or None)
        """
        if self.training:
            raise RuntimeError('forward_tensors is the inference entry point; in training mode call forward(batch)')
        if not marker_pos.is_cuda:
            raise _lib.EmposeError('IterativeErrorFeedback needs GPU tensors; there is no CPU fallback')
        dev = marker_pos.device
        B, F = marker_pos.shape[0], marker_pos.shape[1]
        T = B * F
        f32 = lambda t: None if t is None else t.to(device=dev, dtype=torch.float32).contiguous()
        marker_pos, marker_oris = f32(marker_pos).reshape(B, F, 36), f32(marker_oris).reshape(B, F, 108)
        offset_t, offset_r = f32(offset_t), f32(offset_r)
        if offset_t.shape[0] == 1 and B > 1:
            offset_t, offset_r = offset_t.expand(B, 12, 3).contiguous(), offset_r.expand(B, 12, 3, 3).contiguous()
        assert offset_t.shape == (B, 12, 3) and offset_r.shape == (B, 12, 3, 3)
        marker_masks = f32(marker_masks)
        if seq_lengths is not None:
            seq_lengths = seq_lengths.to(device=dev, dtype=torch.int32).contiguous()
            assert seq_lengths.numel() == B
        lib = _lib.lib()
        with torch.cuda.device(dev):
            handle = self._ensure_handle(dev)
            new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
            io = _lib.LgdIO()
            io.B, io.F = B, F
            io.marker_pos, io.marker_oris = _lib.dptr(marker_pos), _lib.dptr(marker_oris)
            io.offset_t, io.offset_r = _lib.dptr(offset_t), _lib.dptr(offset_r)
            io.marker_masks, io.seq_lengths = _lib.dptr(marker_masks), _lib.dptr(seq_lengths)
            if suppress_mask_value is not None and marker_masks is not None:
                io.suppress_missing, io.mask_value = 1, float(suppress_mask_value)
            out = {'pose': new(B, F, 66), 'shape': new(B, F, 10), 'joints': new(B, F, 66), 'state': None,
                   'hist': None, 'trace': None}
            io.pose_hat, io.shape_hat, io.joints_hat = [_lib.dptr(out[k]) for k in ('pose', 'shape', 'joints')]
            h0 = c0 = None
            if self.rnn_init:
                L, H = self.rnn.num_layers, self.rnn.hidden_size
                if state is not None:
                    h0, c0 = f32(state[0]), f32(state[1])
                    assert h0.shape == (L, B, H) and c0.shape == (L, B, H)
                h_n, c_n = new(L, B, H), new(L, B, H)
                io.h0, io.c0, io.h_n, io.c_n = _lib.dptr(h0), _lib.dptr(c0), _lib.dptr(h_n), _lib.dptr(c_n)
                out['state'] = (h_n, c_n)
            if keep_history:
                n1 = self.N + 1
                hist = {'pose': new(n1, T, 66), 'shape': new(n1, T, 10), 'joints': new(n1, T, 66),
                        'markers': new(n1, T, 36), 'markers_ori': new(n1, T, 108)}
                io.hist_pose, io.hist_shape, io.hist_joints = [_lib.dptr(hist[k]) for k in ('pose', 'shape', 'joints')]
                io.hist_markers, io.hist_markers_ori = _lib.dptr(hist['markers']), _lib.dptr(hist['markers_ori'])
                out['hist'] = hist
            if keep_gradient_trace and self.use_gradient and self.N > 0:
                trace = {'g_pose': new(self.N, T, 66), 'g_shape': new(self.N, T, 10)}
                io.trace_g_pose, io.trace_g_shape = _lib.dptr(trace['g_pose']), _lib.dptr(trace['g_shape'])
                out['trace'] = trace
            need = lib.empose_lgd_workspace_bytes(handle, B, F)
            if self.iter_stream is None:
                if self._workspace is None or self._workspace.numel() < need or self._workspace.device != dev:
                    self._workspace = torch.empty(need, dtype=torch.uint8, device=dev)
                _lib.check(lib.empose_lgd_forward(handle, C.byref(io), _lib.dptr(self._workspace),
                                                  self._workspace.numel(), _lib.current_stream()))
                self.outputs_ready = None
            else:
                side, cur = self.iter_stream, torch.cuda.current_stream(dev)
                if self._pipe is None or self._pipe['dev'] != dev:
                    self._pipe = {'dev': dev, 'turn': 0, 'ws': [None, None], 'free': [None, None]}
                k = self._pipe['turn'] = 1 - self._pipe['turn']
                ws = self._pipe['ws'][k]
                if ws is None or ws.numel() < need:
                    ws = self._pipe['ws'][k] = torch.empty(need, dtype=torch.uint8, device=dev)
                    ws.record_stream(side)
                if self._pipe['free'][k] is not None:
                    cur.wait_event(self._pipe['free'][k])     # the iterations of two chunks ago are done with it
                _lib.check(lib.empose_lgd_forward_phase(handle, C.byref(io), _lib.dptr(ws), ws.numel(),
                                                        C.c_void_p(cur.cuda_stream), 1))
                started = torch.cuda.Event()
                started.record(cur)
                side.wait_event(started)
                # what the side stream reads or writes was allocated for the current stream: keep it from being
                # handed out again before the side stream is done with it
                touched = [offset_t, offset_r, seq_lengths, out['pose'], out['shape'], out['joints']]
                touched += list(out['hist'].values()) if out['hist'] else []
                touched += list(out['trace'].values()) if out['trace'] else []
                for t in touched:
                    if t is not None:
                        t.record_stream(side)
                _lib.check(lib.empose_lgd_forward_phase(handle, C.byref(io), _lib.dptr(ws), ws.numel(),
                                                        C.c_void_p(side.cuda_stream), 2))
                done = torch.cuda.Event()
                done.record(side)
                self._pipe['free'][k] = done
                self.outputs_ready = done
        return out

    # ---- training path (BASELINE configs[4]) -------------------------------------------------------------------
    def residual_gradient(self, pose, shape, inputs_flat, offset_r, offset_t, frame_scale, F):
        """g_pose, g_shape of the in-loop reconstruction energy (reference models.py:560-579) from the HIP kernel."""
        dev, T = pose.device, pose.shape[0]
        lib = _lib.lib()
        f32 = lambda t: t.detach().to(dtype=torch.float32).contiguous()
        pose, shape, tgt = f32(pose), f32(shape), f32(inputs_flat)
        with torch.cuda.device(dev):
            handle = self._ensure_handle(dev)
            new = lambda n: torch.empty(T, n, dtype=torch.float32, device=dev)
            pos, ori, joints, g_pose, g_shape = new(36), new(108), new(66), new(66), new(10)
            nbytes = lib.empose_smpl_workspace_bytes(handle, T)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.empose_smpl_sensors_fwd_bwd(handle, T, F, _lib.dptr(pose), 66, _lib.dptr(shape), 10,
                                                       _lib.dptr(offset_r), _lib.dptr(offset_t), _lib.dptr(tgt),
                                                       tgt.shape[1], _lib.dptr(frame_scale), _lib.dptr(pos),
                                                       _lib.dptr(ori), _lib.dptr(joints), _lib.dptr(g_pose), 66,
                                                       _lib.dptr(g_shape), 10, _lib.dptr(ws), nbytes,
                                                       _lib.current_stream()))
        return g_pose, g_shape

    def _forward_train(self, batch_inputs):
        """
        One window batch WITH an autograd graph (reference models.py:501-609 in training mode).  The SMPL evaluation
        and its reverse are the HIP kernels (custom autograd Function); the LSTM / MLPs run as PyTorch-ROCm ops so
        that autograd provides their backward (train-mode BatchNorm statistics included).  The reference's quirk is
        kept: every in-loop residual gradient is also back-propagated into the parameters that produced the current
        estimate (`E.backward(retain_graph=True)`, models.py:576).
        """
        inputs_ = self.prepare_inputs(batch_inputs)
        if not inputs_.is_cuda:
            raise _lib.EmposeError('IterativeErrorFeedback needs GPU tensors; there is no CPU fallback')
        dev = inputs_.device
        B, F = inputs_.shape[0], inputs_.shape[1]
        T = B * F
        seq_lengths = batch_inputs['seq_lengths'].to(dev)
        masks = batch_inputs['marker_masks']
        offset_r = batch_inputs['offset_r'].to(dev, torch.float32).contiguous()
        offset_t = batch_inputs['offset_t'].to(dev, torch.float32).contiguous()
        live = (torch.arange(F, device=dev)[None, :] < seq_lengths[:, None]).float()
        scale = live * (float(F) / seq_lengths.float())[:, None]
        if masks is not None:
            scale = scale * masks.to(dev).ne(0).all(dim=-1).float()
        scale = scale.reshape(T).contiguous()
        inputs_flat = inputs_.reshape(T, -1)
        if self.rnn_init:
            self.rnn.init_state = self.rnn.final_state
            lstm_out = self.rnn.forward_torch(inputs_, seq_lengths, full_length=getattr(self, 'full_windows', False))
            pose = linear_train(lstm_out, self.pose_net_init).reshape(T, -1)
            shape = linear_train(lstm_out, self.shape_net_init).reshape(T, -1)
        else:
            pose = self.pose_net_init.forward_torch(inputs_flat)
            shape = self.shape_net_init.forward_torch(inputs_flat)

        def single_shape(s):
            return s.reshape(B, F, -1).mean(dim=1, keepdim=True).repeat(1, F, 1).reshape(T, -1)
        if self.shape_avg:
            shape = single_shape(shape)
        hist = {'pose': [], 'shape': [], 'joints': [], 'markers': [], 'markers_ori': []}

        def record(p, s):
            pos, ori, joints = _SmplSensorsFn.apply(self, p, s, offset_r, offset_t, F)
            hist['pose'].append(p)
            hist['shape'].append(s)
            hist['markers'].append(pos)
            hist['markers_ori'].append(ori)
            hist['joints'].append(joints)
        for i in range(self.N):
            feats = [inputs_flat, pose.detach(), shape.detach()]
            if self.use_gradient:
                g_pose, g_shape = self.residual_gradient(pose, shape, inputs_flat, offset_r, offset_t, scale, F)
                # reference quirk (E.backward() inside forward) folded into the one final backward pass
                pose = _InjectGrad.apply(pose, g_pose / float(T))
                shape = _InjectGrad.apply(shape, g_shape / float(T))
                feats += [g_pose, g_shape]
            record(pose, shape)
            x = torch.cat(feats, dim=-1)
            d_pose = self.pose_net_iter.forward_torch(x)
            d_shape = self.shape_net_iter.forward_torch(x)
            if self.shape_avg:
                d_shape = single_shape(d_shape)
            pose = pose + d_pose * self.step_size
            shape = shape + d_shape * self.step_size
        record(pose, shape)
        pose_f = pose.reshape(B, F, -1)
        out = {'pose': pose_f, 'shape': shape.reshape(B, F, -1), 'joints': hist['joints'][-1].reshape(B, F, -1)}
        return out, hist

    def forward(self, batch, window_size=None, is_new_sequence=True):
        if self.rnn_init:
            if is_new_sequence:
                self.rnn.final_state = None
            self.rnn.init_state = self.rnn.final_state
        if self.training or torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and \
                getattr(self, 'differentiable', False):
            self.outputs_ready = None   # everything on the current stream (a caller's iter_stream is not used here)
            return self._forward_with_graph(batch, window_size)
        outs, hists, traces = [], [], []
        # (on the GPU the packing kernel replaces the readings of missing sensors; see RealBatch.get_inputs)
        on_device = bool(batch.seq_lengths.is_cuda)
        for batch_inputs in self.window_generator(batch, window_size=window_size, suppress_on_device=on_device):
            state = None
            if self.rnn_init:
                self.rnn.init_state = self.rnn.final_state
                state = self.rnn.init_state
            res = self.forward_tensors(batch_inputs['marker_pos'], batch_inputs['marker_oris'],
                                       batch_inputs['offset_t'], batch_inputs['offset_r'],
                                       batch_inputs['marker_masks'], batch_inputs['seq_lengths'], state=state,
                                       keep_history=self.keep_history,
                                       keep_gradient_trace=self.keep_gradient_trace,
                                       suppress_mask_value=batch_inputs.get('suppress_mask_value'))
            if self.rnn_init:
                self.rnn.final_state = res['state']
            outs.append(res)
            hists.append(res['hist'])
            traces.append(res['trace'])

        bsz = batch.batch_size
        if self.keep_history:
            # Same shapes as reference models.py:611-629: history entry h is (B, -1, last_dim of the flat tensor).
            def merged(key, inner):
                res = []
                for h in range(self.N + 1):
                    parts = [hw[key][h].reshape(bsz, -1, inner) for hw in hists]
                    res.append(_cat_windows(parts))
                return res
            self.pose_hat_history = merged('pose', 66)
            self.shape_hat_history = merged('shape', 10)
            self.joints_hat_history = merged('joints', 3)
            self.markers_hat_history = merged('markers', 3)
            self.markers_ori_hat_history = merged('markers_ori', 3)
        self.gradient_trace = traces if self.keep_gradient_trace else None
        pose = _cat_windows([o['pose'] for o in outs])
        return {'pose_hat': pose[:, :, 3:], 'root_ori_hat': pose[:, :, :3],
                'shape_hat': _cat_windows([o['shape'] for o in outs]),
                'joints_hat': _cat_windows([o['joints'] for o in outs])}

    def _forward_with_graph(self, batch, window_size):
        from em_pose_amd.nn.train_engine import LgdTrainEngine
        outs, hists = [], []
        windows = list(self.window_generator(batch, window_size=window_size))
        # one window per forward and a released architecture: the hand-written forward / reverse sweep
        # (nn/train_engine.py); anything else builds an autograd graph over the same kernels
        use_engine = len(windows) == 1 and self.training and getattr(self, 'use_train_engine', True) and \
            LgdTrainEngine.supported(self)
        self._engine = None
        for batch_inputs in windows:
            if use_engine:
                self._engine = LgdTrainEngine(self)
                out, hist = self._engine.forward(batch_inputs)
            else:
                out, hist = self._forward_train(batch_inputs)
            outs.append(out)
            hists.append(hist)
        bsz = batch.batch_size

        def merged(key, inner):
            return [_cat_windows([hw[key][h].reshape(bsz, -1, inner) for hw in hists]) for h in range(self.N + 1)]
        self.pose_hat_history = merged('pose', 66)
        self.shape_hat_history = merged('shape', 10)
        self.joints_hat_history = merged('joints', 3)
        self.markers_hat_history = merged('markers', 3)
        self.markers_ori_hat_history = merged('markers_ori', 3)
        pose = _cat_windows([o['pose'] for o in outs])
        return {'pose_hat': pose[:, :, 3:], 'root_ori_hat': pose[:, :, :3],
                'shape_hat': _cat_windows([o['shape'] for o in outs]),
                'joints_hat': _cat_windows([o['joints'] for o in outs])}

    def _select_markers(self, t):
        """(bs, f, 12, d) -> the model's sensors; the index lives on the device (no host copy per call)."""
        if len(self.marker_idxs) == t.shape[2]:
            return t
        cache = getattr(self, '_marker_idx_dev', None)
        if cache is None or cache.device != t.device:
            cache = torch.tensor(self.marker_idxs, dtype=torch.long, device=t.device)
            self._marker_idx_dev = cache
        return t.index_select(2, cache)

    def backward(self, batch, model_out, writer=None, global_step=None, as_tensors=False):
        """Losses of reference models.py:634-688 from the recorded histories; in training mode also
        `total_loss.backward()` (the histories then carry the autograd graph of `_forward_train`).
        `as_tensors=True` returns the loss values as device tensors (no host read-back: capturable in a HIP graph)."""
        if self.pose_hat_history is None:
            raise RuntimeError('backward() needs the histories of the preceding forward() (keep_history=True)')
        if self.training and getattr(self, '_engine', None) is not None:
            # hand-written losses + reverse sweep (nn/train_engine.py); parameter gradients are added to `.grad`
            engine, self._engine = self._engine, None
            total, loss_vals = engine.backward(batch, as_tensors=as_tensors)
            if writer is not None:
                self.log_loss_vals(loss_vals, writer, global_step)
            return total, loss_vals
        bs, f = batch.batch_size, batch.seq_length
        dev = model_out['pose_hat'].device
        inputs_ = self.prepare_inputs(batch.get_inputs()).to(dev)
        markers_in = inputs_[:, :, self.pos_d_start:self.pos_d_end].reshape(bs, f, -1, 3)
        markers_ori_in = inputs_[:, :, self.ori_d_start:self.ori_d_end].reshape(bs, f, -1, 9)
        sl = batch.seq_lengths.to(dev)
        masks = batch.marker_masks.to(dev) if batch.marker_masks is not None else None
        zero = lambda: torch.zeros(1, device=dev)
        rec, shp, pos, fk = zero(), zero(), zero(), zero()
        n_hist = len(self.pose_hat_history)
        for i in range(n_hist):
            pose_hat = self.pose_hat_history[i].reshape(bs, f, -1)
            shape_hat = self.shape_hat_history[i].reshape(bs, f, -1)
            pos += padded_loss(batch.poses.to(dev), pose_hat, self.smpl_loss, sl)
            shp += padded_loss(batch.shapes.to(dev).unsqueeze(1).repeat(1, f, 1), shape_hat, self.smpl_loss, sl)
            if self.do_fk:
                joints_gt = batch.joints_gt.to(dev).reshape(bs, f, -1, 3)
                fk += reconstruction_loss(joints_gt, model_out['joints_hat'].reshape(bs, f, -1, 3), sl, masks)
            m_hat = self._select_markers(self.markers_hat_history[i].reshape(bs, f, -1, 3))
            o_hat = self._select_markers(self.markers_ori_hat_history[i].reshape(bs, f, -1, 9))
            rec += reconstruction_loss(markers_in, m_hat, sl, masks)
            rec += reconstruction_loss(markers_ori_in, o_hat, sl, masks)
        total = (self.pose_weight * pos + self.fk_loss_weight * fk + self.shape_weight * shp + self.r_weight * rec)
        total = total / n_hist
        if as_tensors:
            loss_vals = {'pose': pos / n_hist, 'shape': shp / n_hist, 'reconstruction': rec / n_hist, 'fk': fk / n_hist,
                         'total_loss': total}
            loss_vals = {k: v.detach().reshape(()) for k, v in loss_vals.items()}
        else:
            loss_vals = {'pose': pos.item() / n_hist, 'shape': shp.item() / n_hist,
                         'reconstruction': rec.item() / n_hist, 'fk': fk.item() / n_hist, 'total_loss': total.item()}
        if writer is not None:
            self.log_loss_vals(loss_vals, writer, global_step)
        if self.training:
            total.backward()
        return total, loss_vals
