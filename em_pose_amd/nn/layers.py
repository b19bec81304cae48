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
        y = self.dropout(self.activation_fn(self.batch_norm(self.input_to_hidden(x))))
        for block in self.hidden_layers:
            z = block.layers(y)
            y = y + z if block.use_skip else z
        return self.hidden_to_output(y)

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
        if self.skip_connection:
            raise NotImplementedError('stand-alone MLP.forward does not implement skip connections')
        lib = _lib.lib()
        y = x.contiguous().float()
        lead = y.shape[:-1]
        y = y.reshape(-1, y.shape[-1])
        for lin, bn, act in self.dense_specs():
            w = lin.weight.detach().contiguous()
            if bn is not None:
                scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().contiguous()
                shift = ((lin.bias - bn.running_mean) * scale + bn.bias).detach().contiguous()
            else:
                scale, shift = None, lin.bias.detach().contiguous()
            out = torch.empty(y.shape[0], lin.out_features, device=y.device, dtype=torch.float32)
            slope = float(act.weight.item()) if act is not None else 0.0
            _lib.check(lib.empose_linear_f32(_lib.dptr(y), y.shape[1], _lib.dptr(w), w.shape[1], _lib.dptr(out),
                                             out.shape[1], y.shape[0], lin.out_features, lin.in_features,
                                             _lib.dptr(scale), _lib.dptr(shift), int(act is not None), slope,
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


class RNNLayer(nn.Module):
    """LSTM parameter container + carried state (reference nn/layers.py:80-167)."""

    def __init__(self, input_size, hidden_size, num_layers, output_size=None, bidirectional=False, dropout=0.0,
                 learn_init_state=False):
        super(RNNLayer, self).__init__()
        if bidirectional or learn_init_state or output_size is not None or dropout > 0.0:
            raise NotImplementedError('the LGD init RNN is a plain unidirectional LSTM (reference models.py:427-430)')
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.is_bidirectional = False
        self.num_directions = 1
        self.init_state = None
        self.final_state = None
        self.lstm = nn.LSTM(input_size, hidden_size, num_layers, bidirectional=False)

    def fill_desc(self, desc, keep):
        desc.num_layers, desc.input_size, desc.hidden_size = self.num_layers, self.input_size, self.hidden_size
        for l in range(self.num_layers):
            arrs = [_np32(getattr(self.lstm, '{}_l{}'.format(n, l)))
                    for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
            keep += arrs
            desc.w_ih[l], desc.w_hh[l], desc.b_ih[l], desc.b_hh[l] = [_lib.fptr(a) for a in arrs]

    def forward_torch(self, x, seq_lengths):
        """Training path only: nn.LSTM over packed sequences with the carried state (reference layers.py:133-157)."""
        from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
        packed = pack_padded_sequence(x, seq_lengths.cpu(), batch_first=True, enforce_sorted=False)
        out, self.final_state = self.lstm(packed, self.init_state)
        out, _ = pad_packed_sequence(out, batch_first=True, total_length=x.shape[1])
        return out

    def forward(self, x, seq_lengths):
        raise NotImplementedError('RNNLayer runs inside IterativeErrorFeedback.forward on the HIP path')


class FeedForwardResidualBlock(nn.Module):
    """y = relu(W x + b + x)  (reference nn/layers.py:170-182). Plain torch: only used by the CPU plumbing config."""

    def __init__(self, input_size, output_size):
        super(FeedForwardResidualBlock, self).__init__()
        self.dense = nn.Linear(input_size, output_size)
        self.activate = nn.ReLU()

    def forward(self, x):
        return self.activate(self.dense(x) + x)
