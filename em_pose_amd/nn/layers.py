"""
Parameter containers for the networks of the LGD path, with the reference's module structure so that `state_dict`
keys are identical and released `model.pth` files load unchanged:

  MLP          reference nn/layers.py:46-77   keys input_to_hidden.*, batch_norm.*, activation_fn.weight,
                                              hidden_layers.{h}.layers.{0,4}.* (Linear) .{1,5}.* (BN) .{2,6}.weight
                                              (PReLU), hidden_to_output.*
  LinearLayers reference nn/layers.py:13-43
  RNNLayer     reference nn/layers.py:80-167  keys lstm.{weight_ih,weight_hh,bias_ih,bias_hh}_l{k}

The arithmetic does NOT run through torch.nn: `IterativeErrorFeedback` packs these parameters into the HIP library
(em_pose_amd/nn/models.py) and `MLP.forward` on its own drives the fp32 matrix-core linear kernel layer by layer.
There is no CPU implementation here.
"""
import ctypes as C

import numpy as np
import torch
from torch import nn

from em_pose_amd import _lib


def _np32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


def _gemm_strided(M, N, K, a, a_rs, a_ks, w, w_rs, w_ks, out, bias=None):
    _lib.check(_lib.lib().empose_gemm_strided_f32(M, N, K, _lib.dptr(a), a_rs, a_ks, _lib.dptr(w), w_rs, w_ks,
                                                  _lib.dptr(out), out.shape[1], _lib.dptr(bias), _lib.current_stream()))


class _HipLinearFn(torch.autograd.Function):
    """y = x W^T + b with forward and backward on the split-K GEMM (`empose_gemm_strided_f32`): the training path's
    linear layers at the reference's batch size (12 windows x 32 frames = 384 rows) are small problems for which a
    library GEMM call costs 10-13 us each; dX = dY W and dW = dY^T X read their operands transposed in place."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = x.contiguous(), w.contiguous()
        M, K, N = x.shape[0], x.shape[1], w.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        _gemm_strided(M, N, K, x, K, 1, w, K, 1, y, b)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        M, K, N = x.shape[0], x.shape[1], w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:      # dX[m][k] = sum_n dY[m][n] W[n][k]
            dx = torch.empty(M, K, dtype=torch.float32, device=x.device)
            _gemm_strided(M, K, N, dy, N, 1, w, 1, K, dx)
        if ctx.needs_input_grad[1]:      # dW[n][k] = sum_m dY[m][n] X[m][k]
            dw = torch.empty(N, K, dtype=torch.float32, device=x.device)
            _gemm_strided(N, K, M, dy, 1, N, x, 1, K, dw)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=0)
        return dx, dw, db


# Bumped whenever a kernel updates BatchNorm running statistics through raw pointers (torch's tensor version counters do
# not see such writes); part of the key of the cached inference handle (nn/models.py::_state_key).
BN_STATS_GENERATION = [0]


class _BnPreluFn(torch.autograd.Function):
    """Train-mode BatchNorm1d followed by PReLU (one shared slope), forward and backward as one HIP kernel each
    (`empose_bn_prelu_train_fwd/bwd`); the running statistics are updated in place like torch.nn.BatchNorm1d does."""
    _counters = {}

    @staticmethod
    def forward(ctx, x, gamma, beta, slope, bn):
        x = x.contiguous()
        M, Cn = x.shape
        dev = x.device
        z = torch.empty(M, Cn, dtype=torch.float32, device=dev)
        mean, rstd = torch.empty(Cn, dtype=torch.float32, device=dev), torch.empty(Cn, dtype=torch.float32, device=dev)
        track = bn.track_running_stats and bn.running_mean is not None
        nbytes = _lib.lib().empose_bn_prelu_workspace_bytes(M, Cn)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
        _lib.check(_lib.lib().empose_bn_prelu_train_fwd(
            M, Cn, _lib.dptr(x), Cn, _lib.dptr(gamma), _lib.dptr(beta), _lib.dptr(slope), float(bn.eps),
            float(bn.momentum), _lib.dptr(bn.running_mean) if track else None,
            _lib.dptr(bn.running_var) if track else None, _lib.dptr(bn.num_batches_tracked) if track else None,
            _lib.dptr(z), Cn, _lib.dptr(mean), _lib.dptr(rstd), _lib.dptr(ws), nbytes, _lib.current_stream()))
        if track:
            BN_STATS_GENERATION[0] += 1
        ctx.save_for_backward(x, gamma, beta, slope, mean, rstd)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, gamma, beta, slope, mean, rstd = ctx.saved_tensors
        dz = dz.contiguous()
        M, Cn = x.shape
        dev = x.device
        dx = torch.empty(M, Cn, dtype=torch.float32, device=dev)
        dgamma, dbeta = torch.empty(Cn, dtype=torch.float32, device=dev), torch.empty(Cn, dtype=torch.float32, device=dev)
        dslope = torch.empty(slope.shape, dtype=torch.float32, device=dev)
        partial = torch.empty((Cn + 31) // 32, dtype=torch.float32, device=dev)
        counter = _BnPreluFn._counters.get(dev)
        if counter is None:   # arrival counter of the kernel's last-workgroup reduction: zero once, self re-arming
            counter = _BnPreluFn._counters[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
        nbytes = _lib.lib().empose_bn_prelu_workspace_bytes(M, Cn)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
        _lib.check(_lib.lib().empose_bn_prelu_train_bwd(
            M, Cn, _lib.dptr(x), Cn, _lib.dptr(dz), Cn, _lib.dptr(gamma), _lib.dptr(beta), _lib.dptr(slope),
            _lib.dptr(mean), _lib.dptr(rstd), _lib.dptr(dx), Cn, _lib.dptr(dgamma), _lib.dptr(dbeta), _lib.dptr(dslope),
            _lib.dptr(partial), _lib.dptr(counter), _lib.dptr(ws), nbytes, _lib.current_stream()))
        return dx, dgamma, dbeta, dslope, None


class _HipLinearLargeFn(torch.autograd.Function):
    """y = x W^T + b for any batch: forward on the fp32 matrix-core GEMM (`empose_linear_f32`), dX = dY . W on the same
    kernel against a transposed copy of W (`empose_transpose_f32`), dW = dY^T X and db = column sums of dY in one
    `empose_gemm_atb_f32` (operands read as they lie, reduction over the rows split across workgroups)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = x.contiguous(), w.contiguous()
        M, K, N = x.shape[0], x.shape[1], w.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().empose_linear_f32(_lib.dptr(x), K, _lib.dptr(w), K, _lib.dptr(y), N, M, N, K, None,
                                                _lib.dptr(b), 0, 0.0, _lib.current_stream()))
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        M, K, N = x.shape[0], x.shape[1], w.shape[0]
        dev, lib = x.device, _lib.lib()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = torch.empty(K, N, dtype=torch.float32, device=dev)
            _lib.check(lib.empose_transpose_f32(N, K, _lib.dptr(w), K, _lib.dptr(wt), N, _lib.current_stream()))
            dx = torch.empty(M, K, dtype=torch.float32, device=dev)
            _lib.check(lib.empose_linear_f32(_lib.dptr(dy), N, _lib.dptr(wt), N, _lib.dptr(dx), K, M, K, N, None, None, 0,
                                             0.0, _lib.current_stream()))
        if ctx.needs_input_grad[1]:
            dw = torch.empty(N, K, dtype=torch.float32, device=dev)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = torch.empty(N, dtype=torch.float32, device=dev)
            nbytes = lib.empose_gemm_atb_workspace_bytes(M, N, K)
            ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
            _lib.check(lib.empose_gemm_atb_f32(M, N, K, _lib.dptr(dy), N, _lib.dptr(x), K, _lib.dptr(dw), K,
                                               _lib.dptr(db), _lib.dptr(ws), ws.numel(), _lib.current_stream()))
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=0)
        return dx, dw, db


class _LstmTrainFn(torch.autograd.Function):
    """Stacked uni-directional LSTM over ragged rows with hand-written forward and back-propagation through time
    (`empose_lstm_train_fwd/bwd`): the forward is the inference wavefront kernel saving gates / cell states / incoming
    hidden states; the backward one cell kernel + one recurrent GEMM per step and layer, then the weight gradients as
    three big GEMMs per layer.  Replaces nn.LSTM (MIOpen) + pack/pad on the training path (reference layers.py:133-157)."""

    @staticmethod
    def forward(ctx, x, lens, h0, c0, n_layers, *weights):
        x = x.contiguous()
        B, F, K = x.shape
        H = weights[1].shape[1]
        dev, lib = x.device, _lib.lib()
        p = _lib.LstmParams()
        p.num_layers, p.input_size, p.hidden_size = n_layers, K, H
        ws_ = [w.contiguous() for w in weights]
        for l in range(n_layers):
            p.w_ih[l], p.w_hh[l], p.b_ih[l], p.b_hh[l] = [ws_[4 * l + k].data_ptr() for k in range(4)]
        y = torch.empty(B, F, H, dtype=torch.float32, device=dev)
        h_n = torch.empty(n_layers, B, H, dtype=torch.float32, device=dev)
        c_n = torch.empty(n_layers, B, H, dtype=torch.float32, device=dev)
        save = torch.empty(lib.empose_lstm_train_save_floats(n_layers, B, F, H), dtype=torch.float32, device=dev)
        nbytes = lib.empose_lstm_train_workspace_bytes(C.byref(p), B, F)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.empose_lstm_train_fwd(C.byref(p), B, F, _lib.dptr(x), K, _lib.dptr(lens), _lib.dptr(h0),
                                             _lib.dptr(c0), _lib.dptr(y), _lib.dptr(h_n), _lib.dptr(c_n), _lib.dptr(save),
                                             _lib.dptr(ws), nbytes, _lib.current_stream()))
        ctx.save_for_backward(x, lens, c0, save, *ws_)
        ctx.n_layers = n_layers
        ctx.has_state = h0 is not None
        ctx.mark_non_differentiable(h_n, c_n)
        return y, h_n, c_n

    @staticmethod
    def backward(ctx, dy, _dh, _dc):
        x, lens, c0, save = ctx.saved_tensors[:4]
        weights = ctx.saved_tensors[4:]
        L = ctx.n_layers
        B, F, K = x.shape
        H = weights[1].shape[1]
        dev, lib = x.device, _lib.lib()
        p, g = _lib.LstmParams(), _lib.LstmGrads()
        p.num_layers, p.input_size, p.hidden_size = L, K, H
        grads = [torch.empty_like(w) for w in weights]
        for l in range(L):
            p.w_ih[l], p.w_hh[l], p.b_ih[l], p.b_hh[l] = [weights[4 * l + k].data_ptr() for k in range(4)]
            g.w_ih[l], g.w_hh[l], g.b_ih[l], g.b_hh[l] = [grads[4 * l + k].data_ptr() for k in range(4)]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        # the cotangents of a given initial state (a learned one, reference layers.py:121-131), where autograd wants them
        d_h0 = d_c0 = None
        if ctx.has_state and ctx.needs_input_grad[2]:
            d_h0 = torch.empty(L, B, H, dtype=torch.float32, device=dev)
        if ctx.has_state and ctx.needs_input_grad[3]:
            d_c0 = torch.empty(L, B, H, dtype=torch.float32, device=dev)
        for l in range(L):
            if d_h0 is not None:
                g.d_h0[l] = d_h0[l].data_ptr()
            if d_c0 is not None:
                g.d_c0[l] = d_c0[l].data_ptr()
        nbytes = lib.empose_lstm_train_workspace_bytes(C.byref(p), B, F)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dy = dy.contiguous()
        _lib.check(lib.empose_lstm_train_bwd(C.byref(p), B, F, _lib.dptr(x), K, _lib.dptr(lens), _lib.dptr(c0),
                                             _lib.dptr(save), _lib.dptr(dy), _lib.dptr(dx), C.byref(g), _lib.dptr(ws),
                                             nbytes, _lib.current_stream()))
        return (dx, None, d_h0, d_c0, None) + tuple(grads)


def bn_prelu_train(x, bn, act):
    """`act(bn(x))` for the training path: the fused kernels for a train-mode BatchNorm1d with a momentum and a PReLU
    with one slope on GPU tensors, the torch modules otherwise."""
    if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and isinstance(bn, nn.BatchNorm1d) and bn.training
            and bn.affine and bn.momentum is not None and isinstance(act, nn.PReLU) and act.weight.numel() == 1):
        return _BnPreluFn.apply(x, bn.weight, bn.bias, act.weight, bn)
    return act(bn(x))


FORCE_LARGE_LINEAR = [False]   # dev / fuzz switch: take the large-batch GEMM path (the kernels the training engine uses)


def linear_train(x, lin):
    """`lin(x)` with autograd for the training path: the strided small-problem kernel when the three GEMMs of the layer
    are small (the reference's training batch), the matrix-core GEMM / A^T B kernels otherwise (`_HipLinearLargeFn`);
    plain `lin(x)` only for shapes off the 4-column grid or CPU tensors."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    M, K, N = x2.shape[0], lin.in_features, lin.out_features
    ok = _lib.lib().empose_gemm_strided_applicable
    if x2.is_cuda and x2.dtype == torch.float32:
        if not FORCE_LARGE_LINEAR[0] and ok(M, N) and ok(M, K) and ok(N, K):
            return _HipLinearFn.apply(x2, lin.weight, lin.bias).reshape(lead + (N,))
        if K % 4 == 0 and N % 4 == 0:
            return _HipLinearLargeFn.apply(x2, lin.weight, lin.bias).reshape(lead + (N,))
    return lin(x)


class LinearLayers(nn.Module):
    """`num_layers` x (Linear, BatchNorm1d, PReLU, Dropout) with an optional skip from input to output."""

    def __init__(self, hidden_size, num_layers=2, dropout_p=0.0, use_skip=False, use_batch_norm=True):
        super(LinearLayers, self).__init__()
        self.hidden_size = hidden_size
        self.use_skip = use_skip
        self.use_batch_norm = use_batch_norm
        mods = []
        for _ in range(num_layers):
            mods.append(nn.Linear(hidden_size, hidden_size))
            if use_batch_norm:
                bn = nn.BatchNorm1d(hidden_size)
                nn.init.uniform_(bn.weight)
                mods.append(bn)
            mods.append(nn.PReLU())
            mods.append(nn.Dropout(dropout_p))
        self.layers = nn.Sequential(*mods)

    def dense_specs(self):
        """[(linear, bn or None, prelu)] in execution order."""
        specs, mods = [], list(self.layers)
        step = 4 if self.use_batch_norm else 3
        for k in range(0, len(mods), step):
            lin = mods[k]
            bn = mods[k + 1] if self.use_batch_norm else None
            act = mods[k + (2 if self.use_batch_norm else 1)]
            specs.append((lin, bn, act))
        return specs


class MLP(nn.Module):
    def __init__(self, input_size, output_size, hidden_size, num_layers=2, dropout_p=0.0, skip_connection=False,
                 use_batch_norm=True):
        super(MLP, self).__init__()
        self.use_batch_norm = use_batch_norm
        self.skip_connection = skip_connection
        self.input_to_hidden = nn.Linear(input_size, hidden_size)
        if use_batch_norm:
            self.batch_norm = nn.BatchNorm1d(hidden_size)
            nn.init.uniform_(self.batch_norm.weight)
        else:
            self.batch_norm = nn.Identity()
        self.activation_fn = nn.PReLU()
        self.dropout = nn.Dropout(dropout_p)
        self.hidden_to_output = nn.Linear(hidden_size, output_size)
        self.hidden_layers = nn.Sequential(*[
            LinearLayers(hidden_size, dropout_p=dropout_p, use_batch_norm=use_batch_norm, use_skip=skip_connection)
            for _ in range(num_layers)])

    def dense_specs(self):
        specs = [(self.input_to_hidden, self.batch_norm if self.use_batch_norm else None, self.activation_fn)]
        for block in self.hidden_layers:
            specs.extend(block.dense_specs())
        specs.append((self.hidden_to_output, None, None))
        return specs

    def forward_torch(self, x):
        """The same network as PyTorch-ROCm ops with autograd (training path only: train-mode BatchNorm statistics,
        parameter gradients). Layer order as reference nn/layers.py:69-77."""
        y = self.dropout(bn_prelu_train(linear_train(x, self.input_to_hidden), self.batch_norm, self.activation_fn))
        for block in self.hidden_layers:
            z = y
            mods = list(block.layers)
            step = 4 if block.use_batch_norm else 3      # (Linear, [BatchNorm1d], PReLU, Dropout) per dense layer
            for k in range(0, len(mods), step):
                lin, act, drop = mods[k], mods[k + step - 2], mods[k + step - 1]
                z = linear_train(z, lin)
                z = bn_prelu_train(z, mods[k + 1], act) if block.use_batch_norm else act(z)
                z = drop(z)      # p = 0 in every released configuration; a torch op on the GPU when it is not
            y = y + z if block.use_skip else z
        return linear_train(y, self.hidden_to_output)

    def fill_desc(self, desc, keep):
        """Fill an `_lib.MlpDesc` from the current parameters; host arrays are appended to `keep`."""
        specs = self.dense_specs()
        if len(specs) > _lib.MAX_DENSE:
            raise ValueError('MLP too deep for the HIP library ({} dense layers)'.format(len(specs)))
        desc.n_layers = len(specs)
        desc.skip = int(self.skip_connection)
        for i, (lin, bn, act) in enumerate(specs):
            fill_dense_desc(desc.layers[i], lin, bn, act, keep)

    def forward(self, x):
        """Eval-mode forward on the GPU through the fp32 matrix-core linear kernel (one launch per layer)."""
        if self.training:
            raise NotImplementedError('training mode of MLP is not available on the HIP path yet')
        lib = _lib.lib()
        y = x.contiguous().float()
        lead = y.shape[:-1]
        y = y.reshape(-1, y.shape[-1])
        specs = self.dense_specs()
        block_in = None
        for i, (lin, bn, act) in enumerate(specs):
            w = lin.weight.detach().contiguous()
            if bn is not None:
                scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().contiguous()
                shift = ((lin.bias - bn.running_mean) * scale + bn.bias).detach().contiguous()
            else:
                scale, shift = None, lin.bias.detach().contiguous()
            # hidden block h = dense layers 1+2h, 2+2h; with skip connections its output is block_in + block(block_in)
            in_block = 0 < i < len(specs) - 1
            if in_block and (i - 1) % 2 == 0:
                block_in = y
            resid = block_in if (self.skip_connection and in_block and (i - 1) % 2 == 1) else None
            out = torch.empty(y.shape[0], lin.out_features, device=y.device, dtype=torch.float32)
            slope = float(act.weight.item()) if act is not None else 0.0
            _lib.check(lib.empose_linear_f32_ex(_lib.dptr(y), y.shape[1], _lib.dptr(w), w.shape[1], _lib.dptr(out),
                                                out.shape[1], y.shape[0], lin.out_features, lin.in_features,
                                                _lib.dptr(scale), _lib.dptr(shift), _lib.dptr(resid),
                                                0 if resid is None else resid.shape[1], int(act is not None), slope,
                                                _lib.current_stream()))
            y = out
        return y.reshape(lead + (y.shape[-1],))


def fill_dense_desc(d, lin, bn, act, keep):
    w, b = _np32(lin.weight), _np32(lin.bias)
    keep += [w, b]
    d.in_dim, d.out_dim = lin.in_features, lin.out_features
    d.weight, d.bias = _lib.fptr(w), _lib.fptr(b)
    if bn is not None:
        arrs = [_np32(bn.weight), _np32(bn.bias), _np32(bn.running_mean), _np32(bn.running_var)]
        keep += arrs
        d.bn_weight, d.bn_bias, d.bn_mean, d.bn_var = [_lib.fptr(a) for a in arrs]
        d.bn_eps = float(bn.eps)
    else:
        d.bn_weight = d.bn_bias = d.bn_mean = d.bn_var = None
        d.bn_eps = 0.0
    if act is not None:
        if act.weight.numel() != 1:
            raise ValueError('PReLU with per-channel slopes is not supported')
        d.has_prelu, d.prelu = 1, float(act.weight.detach().cpu().item())
    else:
        d.has_prelu, d.prelu = 0, 0.0


def linear_hip(x2d, lin, resid=None, act=0, slope=0.0):
    """y = act(x W^T + b (+ resid)) on the fp32 matrix-core kernel; x2d (M,K) fp32 contiguous on the GPU."""
    w = lin.weight.detach().contiguous()
    b = lin.bias.detach().contiguous()
    out = torch.empty(x2d.shape[0], lin.out_features, device=x2d.device, dtype=torch.float32)
    _lib.check(_lib.lib().empose_linear_f32_ex(_lib.dptr(x2d), x2d.shape[1], _lib.dptr(w), w.shape[1], _lib.dptr(out),
                                               out.shape[1], x2d.shape[0], lin.out_features, lin.in_features, None,
                                               _lib.dptr(b), _lib.dptr(resid),
                                               0 if resid is None else resid.shape[1], act, slope,
                                               _lib.current_stream()))
    return out


class RNNLayer(nn.Module):
    """
    (Bi)LSTM parameter container + carried state (reference nn/layers.py:80-167).  Inside IterativeErrorFeedback the
    LSTM runs as part of `empose_lgd_forward`; stand-alone (`forward`, used by the BiRNN baseline) it runs through its
    own `empose_rnn_*` handle.
    """

    def __init__(self, input_size, hidden_size, num_layers, output_size=None, bidirectional=False, dropout=0.0,
                 learn_init_state=False):
        super(RNNLayer, self).__init__()
        if bidirectional and learn_init_state:
            raise NotImplementedError('the reference reshapes the learned state to (B, L, H) (layers.py:125-130), which '
                                      'does not fit a bidirectional LSTM')
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.learn_init_state = learn_init_state
        self.is_bidirectional = bidirectional
        self.num_directions = 2 if bidirectional else 1
        # dropout on the inputs (reference layers.py:103-106, 140): active in training mode only, i.e. on `forward_torch`
        self.input_drop = nn.Dropout(p=dropout) if dropout > 0.0 else nn.Identity()
        self.init_state = None
        self.final_state = None
        if self.learn_init_state:
            self.to_init_state_h = nn.Linear(input_size, hidden_size * num_layers * self.num_directions)
            self.to_init_state_c = nn.Linear(input_size, hidden_size * num_layers * self.num_directions)
        self.lstm = nn.LSTM(input_size, hidden_size, num_layers, bidirectional=bidirectional)
        self.to_out = nn.Linear(hidden_size * self.num_directions, output_size) if output_size is not None \
            else nn.Identity()
        self._handle, self._handle_key = None, None

    def _unit_params(self):
        for l in range(self.num_layers):
            for suffix in ([''] if not self.is_bidirectional else ['', '_reverse']):
                yield [getattr(self.lstm, '{}_l{}{}'.format(n, l, suffix))
                       for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]

    def fill_desc(self, desc, keep):
        """Fill the LSTM part of an `_lib.ModelDesc` (uni-directional init RNN of the LGD model)."""
        assert not self.is_bidirectional
        desc.num_layers, desc.input_size, desc.hidden_size = self.num_layers, self.input_size, self.hidden_size
        for l, params in enumerate(self._unit_params()):
            arrs = [_np32(p) for p in params]
            keep += arrs
            desc.w_ih[l], desc.w_hh[l], desc.b_ih[l], desc.b_hh[l] = [_lib.fptr(a) for a in arrs]

    def _ensure_handle(self, device):
        ps = [p for unit in self._unit_params() for p in unit]
        key = (device.index, tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps))
        if self._handle is not None and key == self._handle_key:
            return self._handle
        self.release()
        keep = []
        desc = _lib.RnnDesc()
        desc.num_layers, desc.input_size, desc.hidden_size = self.num_layers, self.input_size, self.hidden_size
        desc.bidirectional = int(self.is_bidirectional)
        for u, params in enumerate(self._unit_params()):
            arrs = [_np32(p) for p in params]
            keep += arrs
            desc.w_ih[u], desc.w_hh[u], desc.b_ih[u], desc.b_hh[u] = [_lib.fptr(a) for a in arrs]
        handle = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().empose_rnn_create(C.byref(desc), C.byref(handle)))
        self._handle, self._handle_key = handle, key
        return handle

    def release(self):
        if getattr(self, '_handle', None) is not None:
            _lib.lib().empose_rnn_destroy(self._handle)
            self._handle, self._handle_key = None, None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def cell_init(self, inputs_):
        """Initial (h_0, c_0); with `learn_init_state` from the first frame (reference layers.py:121-131)."""
        if not self.learn_init_state:
            return self.init_state
        first = inputs_[:, 0].contiguous().float()
        shape = lambda t: t.reshape(-1, self.num_layers, self.hidden_size).transpose(0, 1).contiguous()
        c0 = shape(linear_hip(first, self.to_init_state_c))
        h0 = shape(linear_hip(first, self.to_init_state_h))
        # The reference returns (c0, h0) and nn.LSTM reads the pair as (h_0, c_0): kept as is.
        return c0, h0

    def forward(self, x, seq_lengths):
        """(N, F, input) -> (N, F, directions*hidden) through empose_rnn_fwd; ragged sequences, carried state."""
        if self.training:
            raise NotImplementedError('RNNLayer.forward is the inference path; training uses forward_torch')
        if not x.is_cuda:
            raise _lib.EmposeError('RNNLayer needs GPU tensors; there is no CPU fallback')
        dev = x.device
        B, F = x.shape[0], x.shape[1]
        x = x.contiguous().float()
        self.init_state = self.cell_init(x)
        lens = seq_lengths.to(device=dev, dtype=torch.int32).contiguous()
        U, H = self.num_layers * self.num_directions, self.hidden_size
        lib = _lib.lib()
        with torch.cuda.device(dev):
            handle = self._ensure_handle(dev)
            y = torch.empty(B, F, self.num_directions * H, dtype=torch.float32, device=dev)
            h_n = torch.empty(U, B, H, dtype=torch.float32, device=dev)
            c_n = torch.empty(U, B, H, dtype=torch.float32, device=dev)
            h0 = c0 = None
            if self.init_state is not None:
                h0, c0 = [t.to(device=dev, dtype=torch.float32).contiguous() for t in self.init_state]
                assert h0.shape == (U, B, H) and c0.shape == (U, B, H)
            nbytes = lib.empose_rnn_workspace_bytes(handle, B, F)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.empose_rnn_fwd(handle, B, F, _lib.dptr(x), x.shape[2], _lib.dptr(lens), _lib.dptr(h0),
                                          _lib.dptr(c0), _lib.dptr(y), _lib.dptr(h_n), _lib.dptr(c_n), _lib.dptr(ws),
                                          nbytes, _lib.current_stream()))
        self.final_state = (h_n, c_n)
        if isinstance(self.to_out, nn.Linear):
            y = linear_hip(y.reshape(B * F, -1), self.to_out).reshape(B, F, -1)
        return y

    @staticmethod
    def _reverse_rows(x, lens):
        """Every row's first `lens[b]` frames in reverse order, the padding where it was (what a packed sequence hands to
        the backward direction of a bidirectional LSTM)."""
        F = x.shape[1]
        idx = torch.arange(F, device=x.device).unsqueeze(0)
        n = lens.to(device=x.device, dtype=torch.int64).unsqueeze(1)
        src = torch.where(idx < n, n - 1 - idx, idx)
        return torch.gather(x, 1, src.unsqueeze(-1).expand_as(x))

    def forward_torch(self, x, seq_lengths, full_length=False):
        """Training path with an autograd graph (configurations nn/train_engine.py does not cover, and the baselines):
        the hand-written LSTM forward + back-propagation through time as one autograd Function (reference
        layers.py:133-157).  `full_length=True`: every row spans all F frames (no lengths needed).
        A bidirectional stack is composed from the same Function: per layer one pass over the rows as they are and one
        over the rows reversed within their lengths, outputs concatenated (what nn.LSTM does with a packed sequence)."""
        from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
        x = self.input_drop(x)
        on_hip = x.is_cuda and self.num_layers <= 4 and x.dtype == torch.float32 and x.shape[2] % 4 == 0
        if on_hip and not self.is_bidirectional:
            # hand-written forward + back-propagation through time; no packing, no host round trip, capturable
            lens = None if full_length else seq_lengths.to(device=x.device, dtype=torch.int32).contiguous()
            h0 = c0 = None
            if self.learn_init_state:
                # from the first frame, with a graph (reference layers.py:121-131; it returns (c0, h0) and nn.LSTM reads the
                # pair as (h_0, c_0): kept).  It replaces a carried state, as in the reference.
                first = x[:, 0].contiguous()
                shape = lambda t: t.reshape(-1, self.num_layers, self.hidden_size).transpose(0, 1).contiguous()
                h0, c0 = shape(linear_train(first, self.to_init_state_c)), shape(linear_train(first, self.to_init_state_h))
                self.init_state = (h0, c0)
            elif self.init_state is not None:   # a carried state is a constant (reference models.py:489-492)
                h0, c0 = [t.detach().to(device=x.device, dtype=torch.float32).contiguous() for t in self.init_state]
            weights = [w for unit in self._unit_params() for w in unit]
            y, h_n, c_n = _LstmTrainFn.apply(x, lens, h0, c0, self.num_layers, *weights)
            self.final_state = (h_n, c_n)
            return self._to_out_train(y)
        if on_hip and self.hidden_size % 2 == 0:
            B, F = x.shape[0], x.shape[1]
            lens64 = torch.full((B,), F, dtype=torch.int64, device=x.device) if full_length else \
                seq_lengths.to(device=x.device, dtype=torch.int64)
            lens = lens64.to(torch.int32).contiguous()
            units = list(self._unit_params())
            hs, cs = [], []
            for l in range(self.num_layers):
                outs = []
                for d in range(2):
                    u = 2 * l + d
                    h0 = c0 = None
                    if self.init_state is not None:
                        h0, c0 = [t[u:u + 1].detach().to(device=x.device, dtype=torch.float32).contiguous()
                                  for t in self.init_state]
                    xin = self._reverse_rows(x, lens64) if d == 1 else x
                    y, h_n, c_n = _LstmTrainFn.apply(xin.contiguous(), lens, h0, c0, 1, *units[u])
                    outs.append(self._reverse_rows(y, lens64) if d == 1 else y)
                    hs.append(h_n)
                    cs.append(c_n)
                x = torch.cat(outs, dim=-1)
            self.final_state = (torch.cat(hs, dim=0), torch.cat(cs, dim=0))
            return self._to_out_train(x)
        # fallback (CPU tensors, shapes off the kernels' grid): nn.LSTM over packed sequences, as the reference
        if full_length:
            out, self.final_state = self.lstm(x.transpose(0, 1).contiguous(), self.init_state)
            return self.to_out(out.transpose(0, 1).contiguous())
        packed = pack_padded_sequence(x, seq_lengths.cpu(), batch_first=True, enforce_sorted=False)
        out, self.final_state = self.lstm(packed, self.init_state)
        out, _ = pad_packed_sequence(out, batch_first=True, total_length=x.shape[1])
        return self.to_out(out)

    def _to_out_train(self, y):
        if isinstance(self.to_out, nn.Linear):
            return linear_train(y, self.to_out)
        return y


class FeedForwardResidualBlock(nn.Module):
    """y = relu(W x + b + x)  (reference nn/layers.py:170-182). GPU tensors: one fp32 MFMA launch with the residual
    and the ReLU in the epilogue; CPU tensors: plain torch (BASELINE.json configs[0] is a CPU plumbing check)."""

    def __init__(self, input_size, output_size):
        super(FeedForwardResidualBlock, self).__init__()
        self.dense = nn.Linear(input_size, output_size)
        self.activate = nn.ReLU()

    def forward(self, x):
        if x.is_cuda and not self.training:
            lead = x.shape[:-1]
            x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
            return linear_hip(x2, self.dense, resid=x2, act=2).reshape(lead + (self.dense.out_features,))
        if x.is_cuda and x.dtype == torch.float32:
            # training: the product and its reverse on the HIP GEMMs (autograd Function), residual and ReLU as torch ops
            return self.activate(linear_train(x, self.dense) + x)
        return self.activate(self.dense(x) + x)
