"""
Hand-written training step of the LGD models (BASELINE.json configs[4]).

The reference trains `IterativeErrorFeedback` through torch.autograd: `forward` builds a graph over the LSTM, the N
update-network applications and N + 1 body-model evaluations, `backward` sums the loss terms and calls
`total_loss.backward()` (reference nn/models.py:485-688).  The graph has a fixed, simple shape -- the update networks see
DETACHED inputs (models.py:549-551), so a parameter gradient only needs the cotangent of that application's output, and
estimates are chained by plain additions `pose_{i+1} = pose_i + step * delta_i` -- so this module runs the same
computation as an explicit forward sweep and an explicit reverse sweep over the library's own kernels, without autograd:

  forward   LSTM (empose_lstm_train_fwd) or init MLPs, heads, then per iteration: body model + residual gradient
            (empose_smpl_sensors_fwd_bwd, writing the gradient features straight into the network input rows), the two
            update MLPs in training mode (empose_mlp_train_fwd: GEMMs + train-mode BatchNorm/PReLU kernels), window mean of
            the shape update, additive update; every estimate goes into stacked history buffers.
  backward  all loss terms and their cotangents in one kernel (empose_lgd_losses), then for i = N .. 0: body-model
            vector-Jacobian product (empose_smpl_sensors_vjp), accumulation of the estimate's cotangent (including the
            reference's `E_i.backward()` deposit, models.py:576), update-MLP parameter gradients accumulated over the
            iterations (empose_mlp_train_bwd), finally the heads and back-propagation through time
            (empose_lstm_train_bwd).  Gradients are added to `param.grad` like autograd's AccumulateGrad.

Covers the released configurations (no skip connections, BatchNorm on, dropout 0, one window per forward); anything
else stays on the autograd path of nn/models.py.
"""
import contextlib
import ctypes as C

import torch

from em_pose_amd import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


class _MlpView(object):
    """Device-pointer view of one MLP (reference layers.py:46-77) for empose_mlp_train_*."""

    always_transpose = False     # A/B (scripts/train.py --weight_t): transposed weight copies even where the backward reads W itself

    def __init__(self, mlp):
        self.mlp = mlp
        self.specs = mlp.dense_specs()
        self.n_layers = len(self.specs)
        self.in_dim = self.specs[0][0].in_features
        self.hidden = self.specs[0][0].out_features
        self.out_dim = self.specs[-1][0].out_features
        self.weight_t = None
        self.weight_x3 = None     # pack_forward_x3(): three-piece bf16 copies of the hidden layers' weights (large batches)
        self.weight_t_x3 = None   # prepare_backward(): ... of their transposed copies
        self.save_layout = 0      # fix_layout(): how this step's forward lays out its saved activations

    def fix_layout(self, lib, M):
        """Pin the save-buffer layout of this step to what the library's options select NOW: the backward and weight-gradient
        calls then read the buffer the way the forward wrote it, whatever happens to the options in between."""
        self.save_layout = 0
        p = self.params()
        layout = lib.empose_mlp_train_save_layout(C.byref(p), int(M))
        if layout <= 0:
            _lib.check(layout)
        self.save_layout = layout

    @staticmethod
    def supported(mlp):
        specs = mlp.dense_specs()
        if getattr(mlp, 'skip_connection', False) or not getattr(mlp, 'use_batch_norm', True):
            return False
        if len(specs) > _lib.MAX_DENSE or getattr(mlp.dropout, 'p', 0.0) > 0:
            return False
        h = specs[0][0].out_features
        for lin, bn, act in specs[:-1]:
            if bn is None or act is None or lin.out_features != h or act.weight.numel() != 1 or bn.momentum is None:
                return False
        return specs[0][0].in_features % 4 == 0 and h % 4 == 0

    def params(self):
        p = _lib.MlpParams()
        p.n_layers, p.in_dim, p.hidden, p.out_dim = self.n_layers, self.in_dim, self.hidden, self.out_dim
        for l, (lin, bn, act) in enumerate(self.specs):
            p.weight[l], p.bias[l] = lin.weight.data_ptr(), lin.bias.data_ptr()
            if bn is not None:
                p.bn_weight[l], p.bn_bias[l] = bn.weight.data_ptr(), bn.bias.data_ptr()
                if bn.track_running_stats and bn.running_mean is not None:
                    p.bn_running_mean[l], p.bn_running_var[l] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                    p.bn_num_batches[l] = bn.num_batches_tracked.data_ptr()
                p.prelu[l] = act.weight.data_ptr()
                p.bn_eps, p.bn_momentum = float(bn.eps), float(bn.momentum)
        p.save_layout = self.save_layout
        for l, t in enumerate(self.weight_t or ()):   # transposed once per step (prepare_backward)
            if t is not None:
                p.weight_t[l] = t.data_ptr()
        for name in ('weight_x3', 'weight_t_x3'):     # packed once per step (pack_forward_x3 / prepare_backward)
            for l, t in enumerate(getattr(self, name) or ()):
                if t is not None:
                    getattr(p, name)[l] = t.data_ptr()
        return p

    X3_MIN_ROWS = 1024       # below, the library runs these layers on other kernels (one-launch layers, fp32 tiles)

    def _pack_x3(self, lib, stream, W, ld, n_rows, n_cols):
        nbytes = lib.empose_pack_weight_x3_bytes(int(n_rows), int(n_cols))
        out = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=W.device)
        _lib.check(lib.empose_pack_weight_x3(W.data_ptr(), int(ld), int(n_rows), int(n_cols), out.data_ptr(), stream))
        return out

    def pack_forward_x3(self, lib, stream, M):
        """Three-piece bf16 copies (fragment order) of the hidden layers' weights for the forward products of this step:
        once per step, like the transposed copies -- the weights do not change inside a step.  Large batches only."""
        self.weight_x3 = None
        if M < _MlpView.X3_MIN_ROWS or self.hidden % 64 != 0 or lib.empose_get_option(b'train_x3') == 0:
            return
        self.weight_x3 = []
        for lin, bn, _ in self.specs:
            n_out, n_in = lin.weight.shape
            self.weight_x3.append(self._pack_x3(lib, stream, lin.weight, n_in, n_out, n_in) if bn is not None else None)

    def prepare_backward(self, lib, stream, M=None):
        """W^T of layers 1.. (what dX = dY . W needs on the K-contiguous GEMM): once per step instead of once per
        application of the network -- the weights do not change inside a step.  Not at all when the backward of M rows
        reads W itself (the one-launch layers of small batches)."""
        self.weight_t = None
        if M is not None and not _MlpView.always_transpose:
            p = self.params()
            uses = lib.empose_mlp_train_uses_weight_t(C.byref(p), int(M))
            if uses < 0:
                _lib.check(uses)
            if uses == 0:
                return
        self.weight_t = [None]
        for lin, _, _ in self.specs[1:]:
            n_out, n_in = lin.weight.shape
            ld = (n_out + 3) & ~3
            alloc = torch.empty if ld == n_out else torch.zeros    # only padding columns need the zeros
            t = alloc(n_in, ld, dtype=torch.float32, device=lin.weight.device)
            _lib.check(lib.empose_transpose_f32(n_out, n_in, lin.weight.data_ptr(), n_in, t.data_ptr(), ld, stream))
            self.weight_t.append(t)
        # ... and their three-piece bf16 copies for dA = dY . W on the bf16 matrix cores (the product against W^T
        # [in][ld]: "N" = in features, "K" = ld = out features padded to a multiple of 4 with zero columns)
        self.weight_t_x3 = None
        if M is not None and M >= _MlpView.X3_MIN_ROWS and self.hidden % 64 == 0 and lib.empose_get_option(b'train_x3') != 0:
            self.weight_t_x3 = [None] + [self._pack_x3(lib, stream, t, t.shape[1], t.shape[0], t.shape[1])
                                         for t in self.weight_t[1:]]

    def parameter_list(self):
        out = []
        for lin, bn, act in self.specs:
            out += [lin.weight, lin.bias]
            if bn is not None:
                out += [bn.weight, bn.bias, act.weight]
        return out

    def grads(self, tensors):
        g = _lib.MlpGrads()
        k = 0
        for l, (lin, bn, act) in enumerate(self.specs):
            g.weight[l], g.bias[l] = tensors[k].data_ptr(), tensors[k + 1].data_ptr()
            k += 2
            if bn is not None:
                g.bn_weight[l], g.bn_bias[l], g.prelu[l] = [t.data_ptr() for t in tensors[k:k + 3]]
                k += 3
        return g


class _nothing(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class LgdTrainEngine(object):
    batched_wgrad = True   # dW / db of the update networks once over the N iterations (False: per iteration; A/B and tests)
    # The two update networks of an iteration are independent of each other, forward and backward, and neither their
    # backward nor their weight gradients feed the cotangent chain (the reference detaches the network inputs,
    # models.py:584): with `two_streams` the shape network runs on a side stream beside the pose network, and the
    # weight-gradient products of both (throughput-bound A^T B GEMMs) run there beside the heads' backward and the
    # back-propagation through time of the LSTM (a latency-bound chain of small launches) on the main stream.
    # Pays from a few thousand frames per step on (256 windows: 12.0 -> 10.5 ms); below that the step is a chain of short
    # launches and the cross-stream hand-offs cost more than the overlap gives (12 windows: 4.2 -> 4.9 ms).
    two_streams = True
    two_streams_min_frames = 2048
    # which uses of the side streams are on (A/B: scripts/train.py --streams): 'fwd' the shape network's forward beside the
    # pose network's; 'bwd' its backward; 'bwd3' the pose network's backward on a second side stream, so that the main
    # stream is only the cotangent chain, the heads and back-propagation through time; 'wgrad' the weight-gradient products
    # behind them.  Measured at 256 windows (frames/s): none 684 k, fwd 713 k, bwd 714 k, wgrad 695 k, all 795-810 k.
    side_parts = ('fwd', 'bwd', 'bwd3', 'wgrad')

    def __init__(self, net):
        self.net = net
        self.ctx = None
        self._side_streams = {}
        self._use_side = False
        self._held = []
        self._forked = set()      # side streams forked and not yet joined

    # ---- the side stream ------------------------------------------------------------------------------------------
    def _side(self, k=0):
        st = self._side_streams.get(k)
        if st is None or st.device != self.dev:
            st = self._side_streams[k] = torch.cuda.Stream(device=self.dev)
        return st

    def _fork(self, k=0):
        """Side stream k continues from what the main stream has enqueued so far."""
        if self._use_side:
            self._side(k).wait_stream(torch.cuda.current_stream(self.dev))
            self._forked.add(k)

    def _join(self, k=0):
        """The main stream continues after what side stream k has enqueued so far.  Only a stream that was forked since its
        last join is waited for: under HIP-graph capture a wait on an event of a stream that never joined the capture is an
        isolation error, and a step whose options leave a stream unused must not touch it."""
        if self._use_side and k in self._forked:
            torch.cuda.current_stream(self.dev).wait_stream(self._side(k))
            self._forked.discard(k)
        if not self._forked:
            self._held = []       # nothing is running beside the main stream: its workspaces may go back to the allocator

    def _hold(self, t):
        """A main-stream workspace must not go back to the allocator between a fork and the next join: the block would be
        handed to the next main-pool allocation, and that tensor may be written by the side stream (which forked before
        the main-stream kernels that still use the workspace were enqueued)."""
        if self._use_side and self._forked:
            self._held.append(t)
        return t

    @contextlib.contextmanager
    def _on_side(self, k=0):
        """Launches (and transient allocations: workspaces, freed right after the launch, must belong to the stream that
        uses them) inside go to the side stream.  Long-lived tensors are allocated outside, on the main stream's pool,
        and freed only after a join."""
        if not self._use_side:
            yield
            return
        main_raw = self.stream
        with torch.cuda.stream(self._side(k)):
            self.stream = _lib.current_stream()
            try:
                yield
            finally:
                self.stream = main_raw

    @staticmethod
    def supported(net):
        if net.skip_connections or getattr(net.config, 'm_dropout_hidden', 0.0) > 0 or \
                getattr(net.config, 'm_dropout', 0.0) > 0:
            return False
        mlps = [net.pose_net_iter, net.shape_net_iter] + ([] if net.rnn_init else [net.pose_net_init, net.shape_net_init])
        if not all(_MlpView.supported(m) for m in mlps):
            return False
        if net.rnn_init and (net.rnn.is_bidirectional or net.rnn.num_layers > 4 or net.rnn.learn_init_state):
            return False
        return net.input_size % 4 == 0 and net.input_iter_size % 4 == 0

    # ---- small helpers over the C ABI -------------------------------------------------------------------------
    def _axpby(self, rows, cols, alpha, x, ldx, beta, y, ldy, out, ldo):
        _lib.check(self.lib.empose_axpby2d(rows, cols, alpha, x, ldx, beta, y, ldy, out, ldo, self.stream))

    def _mlp_fwd(self, view, x, ldx, out, ld_out, M, side=None):
        p = view.params()
        save = self.new(self.lib.empose_mlp_train_save_floats(C.byref(p), M))
        nbytes = self.lib.empose_mlp_train_workspace_bytes(C.byref(p), M)
        with (self._on_side(side) if side is not None else _nothing()):
            ws = self.ws(nbytes) if side is not None else self._hold(self.ws(nbytes))
            _lib.check(self.lib.empose_mlp_train_fwd(C.byref(p), M, x, ldx, out, ld_out, save.data_ptr(), ws.data_ptr(),
                                                     nbytes, self.stream))
        from em_pose_amd.nn import layers as _layers
        _layers.BN_STATS_GENERATION[0] += 1
        return save

    def _mlp_fwd_pair(self, views, x, ldx, outs, ld_outs, M):
        """Both update networks of an iteration in one call (empose_mlp_train_fwd_pair: at the reference's training batch
        every layer of both is one launch); returns their save buffers."""
        ps = [v.params() for v in views]
        saves = [self.new(self.lib.empose_mlp_train_save_floats(C.byref(p), M)) for p in ps]
        nbytes = self.lib.empose_mlp_train_pair_workspace_bytes(C.byref(ps[0]), C.byref(ps[1]), M)
        ws = self._hold(self.ws(nbytes))
        _lib.check(self.lib.empose_mlp_train_fwd_pair(C.byref(ps[0]), C.byref(ps[1]), M, x, ldx, outs[0], ld_outs[0],
                                                      outs[1], ld_outs[1], saves[0].data_ptr(), saves[1].data_ptr(),
                                                      ws.data_ptr(), nbytes, self.stream))
        from em_pose_amd.nn import layers as _layers
        _layers.BN_STATS_GENERATION[0] += 2
        return saves

    def _mlp_bwd_deferred_pair(self, views, x, ldx, d_outs, ld_douts, saves, grads, accumulate, M, stashes):
        ps = [v.params() for v in views]
        gs = [v.grads(g) for v, g in zip(views, grads)]
        nbytes = self.lib.empose_mlp_train_pair_workspace_bytes(C.byref(ps[0]), C.byref(ps[1]), M)
        ws = self._hold(self.ws(nbytes))
        _lib.check(self.lib.empose_mlp_train_bwd_deferred_pair(
            C.byref(ps[0]), C.byref(ps[1]), M, x, ldx, d_outs[0], ld_douts[0], d_outs[1], ld_douts[1],
            saves[0].data_ptr(), saves[1].data_ptr(), C.byref(gs[0]), C.byref(gs[1]), int(accumulate),
            stashes[0].data_ptr(), stashes[1].data_ptr(), ws.data_ptr(), nbytes, self.stream))
        return stashes

    def _mlp_bwd(self, view, x, ldx, d_out, ld_dout, save, grads, accumulate, M):
        p = view.params()
        g = view.grads(grads)
        nbytes = self.lib.empose_mlp_train_workspace_bytes(C.byref(p), M)
        ws = self.ws(nbytes)
        _lib.check(self.lib.empose_mlp_train_bwd(C.byref(p), M, x, ldx, d_out, ld_dout, save.data_ptr(), C.byref(g),
                                                 int(accumulate), ws.data_ptr(), nbytes, self.stream))

    def _new_stash(self, view, M):
        """Stash of one deferred application + the address of its d_out slot (row stride (out_dim + 3) & ~3): the
        cotangent kernel writes the output cotangent there directly, so the deferred backward copies nothing."""
        p = view.params()
        stash = self.new(self.lib.empose_mlp_train_stash_floats(C.byref(p), M))
        return stash, stash.data_ptr() + 4 * M * (view.n_layers - 1) * view.hidden

    def _mlp_bwd_deferred(self, view, x, ldx, d_out, ld_dout, save, grads, accumulate, M, stash=None, side=None):
        """Backward of one application that keeps the layer cotangents instead of forming dW / db; returns the stash."""
        p = view.params()
        g = view.grads(grads)
        if stash is None:
            stash = self.new(self.lib.empose_mlp_train_stash_floats(C.byref(p), M))
        nbytes = self.lib.empose_mlp_train_workspace_bytes(C.byref(p), M)
        with (self._on_side(side) if side is not None else _nothing()):
            ws = self.ws(nbytes) if side is not None else self._hold(self.ws(nbytes))
            _lib.check(self.lib.empose_mlp_train_bwd_deferred(C.byref(p), M, x, ldx, d_out, ld_dout, save.data_ptr(),
                                                              C.byref(g), int(accumulate), stash.data_ptr(),
                                                              ws.data_ptr(), nbytes, self.stream))
        return stash

    def _mlp_wgrad(self, view, xs, ldx, saves, stashes, grads, M):
        """dW, db of one network over all its applications: one A^T B product per layer (empose_mlp_train_wgrad)."""
        p = view.params()
        g = view.grads(grads)
        n = len(xs)
        arr = lambda ptrs: (C.c_void_p * n)(*ptrs)
        nbytes = self.lib.empose_mlp_train_wgrad_workspace_bytes(C.byref(p), n, M)
        ws = self.ws(nbytes)
        _lib.check(self.lib.empose_mlp_train_wgrad(C.byref(p), n, M, arr(xs), ldx, arr([t.data_ptr() for t in saves]),
                                                   arr([t.data_ptr() for t in stashes]), C.byref(g), 0, ws.data_ptr(),
                                                   nbytes, self.stream))

    def ws(self, nbytes):
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=self.dev)

    # ---- where gradients go -------------------------------------------------------------------------------------
    @staticmethod
    def gradient_order(net):
        """The parameters in the order in which the reverse sweep finishes their gradients: update networks, then what
        produced the initial estimate (heads, LSTM last).  `helpers.distributed.GradientBuckets` lays its flat buckets
        out in this order so that the first buckets can be all-reduced while the rest is still being computed."""
        order = []
        for mlp in (net.pose_net_iter, net.shape_net_iter):
            order += _MlpView(mlp).parameter_list()
        if net.rnn_init:
            order += [net.pose_net_init.weight, net.pose_net_init.bias, net.shape_net_init.weight,
                      net.shape_net_init.bias]
            order += [w for unit in net.rnn._unit_params() for w in unit]
        else:
            for mlp in (net.pose_net_init, net.shape_net_init):
                order += _MlpView(mlp).parameter_list()
        seen, out = set(), []
        for p in order:
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                out.append(p)
        return out

    def _grad_like(self, p):
        """Where the kernels write p's gradient: p's slice of a persistent flat bucket when a gradient sink is attached
        and nothing has been accumulated into p.grad yet (then the slice simply BECOMES p.grad), else a fresh tensor."""
        sink = getattr(self.net, '_grad_sink', None)
        if sink is not None and p.grad is None and p.requires_grad:
            view = sink.view_of(p)
            if view is not None:
                return view
        return torch.empty_like(p)

    def _deposit(self, named):
        """autograd's AccumulateGrad for a finished group of parameters, then tell the sink that they are final."""
        sink = getattr(self.net, '_grad_sink', None)
        for p_, g_ in named:
            if not p_.requires_grad:
                continue
            if p_.grad is None:
                p_.grad = g_
            elif p_.grad.data_ptr() != g_.data_ptr():
                self._axpby(1, g_.numel(), 1.0, g_.data_ptr(), g_.numel(), 1.0, p_.grad.data_ptr(), g_.numel(),
                            p_.grad.data_ptr(), g_.numel())
            if sink is not None:
                sink.stage(p_)

    def new(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.dev)

    # ---- forward ------------------------------------------------------------------------------------------------
    def forward(self, batch_inputs):
        net = self.net
        if not batch_inputs['marker_pos'].is_cuda:
            raise _lib.EmposeError('IterativeErrorFeedback needs GPU tensors; there is no CPU fallback')
        dev = self.dev = batch_inputs['marker_pos'].device
        lib = self.lib = _lib.lib()
        self.stream = _lib.current_stream()
        seq_lengths = batch_inputs['seq_lengths'].to(dev)
        lens32 = seq_lengths.to(torch.int32).contiguous()
        masks = batch_inputs['marker_masks']
        masks = None if masks is None else masks.to(dev, torch.float32).contiguous()
        # network input rows + the per-frame weight of the in-loop residual (reference loss.py:36-39 times the B * F
        # rescale of models.py:578-579) in one launch
        from em_pose_amd.nn.models import pack_sensor_inputs
        inputs_, scale = pack_sensor_inputs(batch_inputs['marker_pos'], batch_inputs['marker_oris'], net.marker_idxs,
                                            masks, lens32, want_frame_weight=True)
        B, F = inputs_.shape[0], inputs_.shape[1]
        T, N, s = B * F, net.N, float(net.step_size)
        self._use_side = bool(self.two_streams and T >= self.two_streams_min_frames)
        d_in, d_x = net.input_size, net.input_iter_size
        x0 = inputs_.reshape(T, d_in)
        masks = None if masks is None else masks.reshape(T, 12)
        offset_r = batch_inputs['offset_r'].to(dev, torch.float32).contiguous()
        offset_t = batch_inputs['offset_t'].to(dev, torch.float32).contiguous()

        ctx = self.ctx = {'B': B, 'F': F, 'x0': x0, 'lens32': lens32, 'masks': masks, 'offset_r': offset_r,
                          'offset_t': offset_t}
        pose_hist, shape_hist = self.new(N + 1, T, 66), self.new(N + 1, T, 10)
        markers_hist, ori_hist = self.new(N + 1, T, 36), self.new(N + 1, T, 108)
        joints_hist = self.new(N + 1, T, 66)
        tmp10 = self.new(T, 10)
        with torch.cuda.device(dev):
            smpl_h = net._ensure_smpl_handle(dev)
            if net.rnn_init:
                rnn = net.rnn
                rnn.init_state = rnn.final_state
                L, H = rnn.num_layers, rnn.hidden_size
                weights = [w for unit in rnn._unit_params() for w in unit]
                p = _lib.LstmParams()
                p.num_layers, p.input_size, p.hidden_size = L, d_in, H
                for l in range(L):
                    p.w_ih[l], p.w_hh[l], p.b_ih[l], p.b_hh[l] = [weights[4 * l + k].data_ptr() for k in range(4)]
                h0 = c0 = None
                if rnn.init_state is not None:
                    h0, c0 = [t.detach().to(dev, torch.float32).contiguous() for t in rnn.init_state]
                y = self.new(T, H)
                h_n, c_n = self.new(L, B, H), self.new(L, B, H)
                save = self.new(lib.empose_lstm_train_save_floats(L, B, F, H))
                nbytes = lib.empose_lstm_train_workspace_bytes(C.byref(p), B, F)
                ws = self.ws(nbytes)
                _lib.check(lib.empose_lstm_train_fwd(C.byref(p), B, F, x0.data_ptr(), d_in, lens32.data_ptr(), _ptr(h0),
                                                     _ptr(c0), y.data_ptr(), h_n.data_ptr(), c_n.data_ptr(),
                                                     save.data_ptr(), ws.data_ptr(), nbytes, self.stream))
                rnn.final_state = (h_n, c_n)
                ctx.update({'lstm_save': save, 'y': y, 'c0': c0})
                for lin, out, ld in ((net.pose_net_init, pose_hist[0], 66), (net.shape_net_init, tmp10, 10)):
                    _lib.check(lib.empose_linear_f32(y.data_ptr(), H, lin.weight.data_ptr(), H, out.data_ptr(), ld, T,
                                                     lin.out_features, H, None, lin.bias.data_ptr(), 0, 0.0, self.stream))
            else:
                ctx['init_views'] = (_MlpView(net.pose_net_init), _MlpView(net.shape_net_init))
                for v in ctx['init_views']:
                    v.fix_layout(self.lib, T)
                    v.pack_forward_x3(self.lib, self.stream, T)
                ctx['init_saves'] = (self._mlp_fwd(ctx['init_views'][0], x0.data_ptr(), d_in, pose_hist[0].data_ptr(), 66, T),
                                     self._mlp_fwd(ctx['init_views'][1], x0.data_ptr(), d_in, tmp10.data_ptr(), 10, T))
            if net.shape_avg:
                _lib.check(lib.empose_window_mean(T, F, 10, tmp10.data_ptr(), 10, shape_hist[0].data_ptr(), 10, self.stream))
            else:
                self._axpby(T, 10, 1.0, tmp10.data_ptr(), 10, 0.0, None, 0, shape_hist[0].data_ptr(), 10)

            views = (_MlpView(net.pose_net_iter), _MlpView(net.shape_net_iter))
            for v in views:
                v.fix_layout(self.lib, T)
                v.pack_forward_x3(self.lib, self.stream, T)
            X = self.new(max(N, 1), T, d_x)
            dp, ds = self.new(T, 66), self.new(T, 10)
            saves = []
            nb_smpl = lib.empose_smpl_workspace_bytes(smpl_h, T)
            ws_smpl = self.ws(nb_smpl)
            for i in range(N + 1):
                want_g = i < N and net.use_gradient
                Xi = X[i] if i < N else None
                _lib.check(lib.empose_smpl_sensors_fwd_bwd(
                    smpl_h, T, F, pose_hist[i].data_ptr(), 66, shape_hist[i].data_ptr(), 10, offset_r.data_ptr(),
                    offset_t.data_ptr(), x0.data_ptr() if want_g else None, d_in, scale.data_ptr() if want_g else None,
                    markers_hist[i].data_ptr(), ori_hist[i].data_ptr(), joints_hist[i].data_ptr(),
                    Xi[:, d_in + 76:].data_ptr() if want_g else None, d_x,
                    Xi[:, d_in + 142:].data_ptr() if want_g else None, d_x, ws_smpl.data_ptr(), nb_smpl, self.stream))
                if i == N:
                    break
                # network input rows [x0 | pose_i | shape_i | g_pose | g_shape] (the gradients are already there)
                _lib.check(lib.empose_lgd_assemble_inputs(T, d_in, x0.data_ptr(), d_in, pose_hist[i].data_ptr(),
                                                          shape_hist[i].data_ptr(), Xi.data_ptr(), d_x, self.stream))
                side_fwd = self._use_side and 'fwd' in self.side_parts
                if side_fwd:
                    self._fork()                               # the two networks side by side
                    sp = self._mlp_fwd(views[0], Xi.data_ptr(), d_x, dp.data_ptr(), 66, T)
                    ss = self._mlp_fwd(views[1], Xi.data_ptr(), d_x, tmp10.data_ptr(), 10, T, side=0)
                    self._join()
                else:                                          # one stream: both networks per call (paired launches)
                    sp, ss = self._mlp_fwd_pair(views, Xi.data_ptr(), d_x, (dp.data_ptr(), tmp10.data_ptr()), (66, 10), T)
                saves.append((sp, ss))
                # pose_{i+1} = pose_i + s dp, shape_{i+1} = shape_i + s (window mean of) ds
                _lib.check(lib.empose_lgd_additive_update(B, F, s, int(bool(net.shape_avg)), pose_hist[i].data_ptr(),
                                                          dp.data_ptr(), shape_hist[i].data_ptr(), tmp10.data_ptr(),
                                                          pose_hist[i + 1].data_ptr(), shape_hist[i + 1].data_ptr(),
                                                          self.stream))
        ctx.update({'pose_hist': pose_hist, 'shape_hist': shape_hist, 'markers_hist': markers_hist, 'ori_hist': ori_hist,
                    'joints_hist': joints_hist, 'X': X, 'views': views, 'saves': saves, 'smpl_h': smpl_h})
        hist = {'pose': list(pose_hist), 'shape': list(shape_hist), 'joints': list(joints_hist),
                'markers': list(markers_hist), 'markers_ori': list(ori_hist)}
        out = {'pose': pose_hist[N].reshape(B, F, 66), 'shape': shape_hist[N].reshape(B, F, 10),
               'joints': joints_hist[N].reshape(B, F, 66)}
        return out, hist

    # ---- backward -----------------------------------------------------------------------------------------------
    def backward(self, batch, as_tensors=False):
        """Loss values + parameter gradients (added to `.grad`).  :return: (total loss tensor, loss_vals dict)"""
        net, ctx = self.net, self.ctx
        if ctx is None:
            raise RuntimeError('backward() needs the preceding training-mode forward()')
        lib, dev = self.lib, self.dev
        B, F = ctx['B'], ctx['F']
        T, N, s = B * F, net.N, float(net.step_size)
        d_in, d_x = net.input_size, net.input_iter_size
        f32 = lambda t: t.to(dev, torch.float32).contiguous()
        io = _lib.LossIO()
        io.B, io.F, io.n_hist, io.n_markers = B, F, N + 1, net.n_markers
        for k, v in enumerate(net.marker_idxs):
            io.marker_idx[k] = v
        pose_gt, shape_gt = f32(batch.poses).reshape(T, 66), f32(batch.shapes)
        joints_gt = f32(batch.joints_gt).reshape(T, 66) if net.do_fk else None
        d_pose, d_shape = self.new(N + 1, T, 66), self.new(N + 1, T, 10)
        d_mark, d_ori, d_joints = self.new(N + 1, T, 36), self.new(N + 1, T, 108), self.new(T, 66)
        loss_vals = self.new(5)
        io.pose_hist, io.shape_hist = ctx['pose_hist'].data_ptr(), ctx['shape_hist'].data_ptr()
        io.markers_hist, io.markers_ori_hist = ctx['markers_hist'].data_ptr(), ctx['ori_hist'].data_ptr()
        io.joints_final = ctx['joints_hist'][N].data_ptr()
        io.pose_gt, io.shape_gt, io.joints_gt = pose_gt.data_ptr(), shape_gt.data_ptr(), _ptr(joints_gt)
        io.inputs, io.ld_inputs = ctx['x0'].data_ptr(), d_in
        io.seq_lengths, io.marker_masks = ctx['lens32'].data_ptr(), _ptr(ctx['masks'])
        io.w_pose, io.w_shape = float(net.pose_weight), float(net.shape_weight)
        io.w_fk, io.w_rec = float(net.fk_loss_weight) if net.do_fk else 0.0, float(net.r_weight)
        io.d_pose, io.d_shape, io.d_markers, io.d_markers_ori = [t.data_ptr() for t in (d_pose, d_shape, d_mark, d_ori)]
        io.d_joints, io.loss_vals = d_joints.data_ptr(), loss_vals.data_ptr()
        with torch.cuda.device(dev):
            self.stream = _lib.current_stream()
            nbytes = lib.empose_lgd_losses_workspace_bytes(B, F, N + 1)
            ws = self.ws(nbytes)
            _lib.check(lib.empose_lgd_losses(C.byref(io), ws.data_ptr(), nbytes, self.stream))

            smpl_h = ctx['smpl_h']
            nb_vjp = lib.empose_smpl_vjp_workspace_bytes(smpl_h, T)
            ws_vjp = self.ws(nb_vjp)
            Dp, Ds = self.new(T, 66), self.new(T, 10)
            vp, vs = self.new(T, 66), self.new(T, 10)
            dpad = torch.zeros(T, 68, dtype=torch.float32, device=dev)     # zero padding columns for the GEMMs
            dspad = torch.zeros(T, 12, dtype=torch.float32, device=dev)
            tmp10 = self.new(T, 10)
            views = ctx['views']
            grads = [[self._grad_like(p) for p in v.parameter_list()] for v in views]
            X = ctx['X']
            deferred = self.batched_wgrad and 1 <= N <= 8   # (row counts off the 32-row grid: per application inside)
            for v in views:
                v.prepare_backward(lib, self.stream, T)
            pend = ([], [])
            for i in range(N, -1, -1):
                _lib.check(lib.empose_smpl_sensors_vjp(
                    smpl_h, T, F, ctx['pose_hist'][i].data_ptr(), 66, ctx['shape_hist'][i].data_ptr(), 10,
                    ctx['offset_r'].data_ptr(), ctx['offset_t'].data_ptr(), d_mark[i].data_ptr(), d_ori[i].data_ptr(),
                    d_joints.data_ptr() if i == N else None, vp.data_ptr(), vs.data_ptr(), ws_vjp.data_ptr(), nb_vjp,
                    self.stream))
                # running cotangents of the estimates (loss terms + body-model VJP + the reference's in-forward
                # `E_i.backward()` deposit, models.py:576: dE_i/d(pose_i) = g_i / (B F) flows into everything that produced
                # pose_i) and, for i > 0, the zero-padded cotangents of the update networks' outputs of iteration i - 1
                deposit = i < N and net.use_gradient
                dp_ptr, ds_ptr = dpad.data_ptr(), dspad.data_ptr()
                if deferred and i > 0:   # straight into the d_out slots of this application's stashes
                    (st_p, dp_ptr), (st_s, ds_ptr) = self._new_stash(views[0], T), self._new_stash(views[1], T)
                _lib.check(lib.empose_lgd_cotangent_step(
                    B, F, int(i == N), d_pose[i].data_ptr(), d_shape[i].data_ptr(), vp.data_ptr(), vs.data_ptr(),
                    X[i][:, d_in + 76:].data_ptr() if deposit else None, d_x,
                    X[i][:, d_in + 142:].data_ptr() if deposit else None, d_x, Dp.data_ptr(), Ds.data_ptr(), s,
                    int(bool(net.shape_avg)), dp_ptr if i > 0 else None, ds_ptr if i > 0 else None,
                    self.stream))
                if i == 0:
                    break
                sp, ss = ctx['saves'][i - 1]
                acc = i < N
                if deferred:
                    # weight gradients once over all N applications (one A^T B per layer instead of N).  Nothing below
                    # reads what these two calls write until the weight-gradient products: the shape network's backward
                    # of every iteration trails on the side stream, in order, without a join
                    side_bwd = self._use_side and 'bwd' in self.side_parts
                    side_pose = 1 if (side_bwd and 'bwd3' in self.side_parts) else None   # third stream: the pose net too
                    if side_bwd:
                        self._fork(0)
                    if side_pose is not None:
                        self._fork(1)
                    if not side_bwd:                           # one stream: both networks per call (paired launches)
                        self._mlp_bwd_deferred_pair(views, X[i - 1].data_ptr(), d_x, (dp_ptr, ds_ptr), (68, 12), (sp, ss),
                                                    grads, acc, T, (st_p, st_s))
                        pend[0].append((X[i - 1].data_ptr(), sp, st_p))
                        pend[1].append((X[i - 1].data_ptr(), ss, st_s))
                        continue
                    pend[0].append((X[i - 1].data_ptr(), sp,
                                    self._mlp_bwd_deferred(views[0], X[i - 1].data_ptr(), d_x, dp_ptr, 68, sp,
                                                           grads[0], acc, T, stash=st_p, side=side_pose)))
                    pend[1].append((X[i - 1].data_ptr(), ss,
                                    self._mlp_bwd_deferred(views[1], X[i - 1].data_ptr(), d_x, ds_ptr, 12, ss,
                                                           grads[1], acc, T, stash=st_s, side=0 if side_bwd else None)))
                else:
                    self._mlp_bwd(views[0], X[i - 1].data_ptr(), d_x, dpad.data_ptr(), 68, sp, grads[0], acc, T)
                    self._mlp_bwd(views[1], X[i - 1].data_ptr(), d_x, dspad.data_ptr(), 12, ss, grads[1], acc, T)
            # Both networks' weight gradients on the side stream (after the pose network's backward, which ran on the main
            # stream), beside the initial estimate's backward below on the main stream.
            # (per-application gradients -- `deferred` off -- were formed on the main stream: nothing to move aside)
            side_w = deferred and 'wgrad' in self.side_parts
            three = side_w and 'bwd3' in self.side_parts and 'bwd' in self.side_parts
            if deferred and not side_w:     # (A/B: backward on side streams but the products on the main one)
                self._join(0)
                self._join(1)
            if side_w and not three:
                self._fork(0)
            for k in (0, 1):
                # (three streams: each network's products follow its own backward on its own stream, no fork needed)
                where = (1 - k) if three else 0
                with (self._on_side(where) if side_w else _nothing()):
                    if pend[k]:
                        self._mlp_wgrad(views[k], [q[0] for q in pend[k]], d_x, [q[1] for q in pend[k]],
                                        [q[2] for q in pend[k]], grads[k], T)
                    if N > 0:   # final: a gradient sink may start averaging them while the rest of the sweep runs
                        self._deposit(list(zip(views[k].parameter_list(), grads[k])))
            # ---- initial estimate
            self._axpby(T, 66, 1.0, Dp.data_ptr(), 66, 0.0, None, 0, dpad.data_ptr(), 68)
            if net.shape_avg:
                _lib.check(lib.empose_window_mean(T, F, 10, Ds.data_ptr(), 10, tmp10.data_ptr(), 10, self.stream))
                self._axpby(T, 10, 1.0, tmp10.data_ptr(), 10, 0.0, None, 0, dspad.data_ptr(), 12)
            else:
                self._axpby(T, 10, 1.0, Ds.data_ptr(), 10, 0.0, None, 0, dspad.data_ptr(), 12)
            named = []
            if net.rnn_init:
                rnn, y = net.rnn, ctx['y']
                H, L = rnn.hidden_size, rnn.num_layers
                dy = self.new(T, H)
                first = True
                for lin, dpd, ld, n_out in ((net.pose_net_init, dpad, 68, 66), (net.shape_net_init, dspad, 12, 10)):
                    gw, gb = self._grad_like(lin.weight), self._grad_like(lin.bias)
                    nb = lib.empose_gemm_atb_workspace_bytes(T, n_out, H)
                    wsa = self.ws(nb)
                    _lib.check(lib.empose_gemm_atb_f32(T, n_out, H, dpd.data_ptr(), ld, y.data_ptr(), H, gw.data_ptr(), H,
                                                       gb.data_ptr(), wsa.data_ptr(), wsa.numel(), self.stream))
                    named += [(lin.weight, gw), (lin.bias, gb)]
                    wt = torch.zeros(H, ld, dtype=torch.float32, device=dev)
                    _lib.check(lib.empose_transpose_f32(n_out, H, lin.weight.data_ptr(), H, wt.data_ptr(), ld, self.stream))
                    _lib.check(lib.empose_linear_f32_ex(dpd.data_ptr(), ld, wt.data_ptr(), ld, dy.data_ptr(), H, T, H, ld,
                                                        None, None, None if first else dy.data_ptr(), H, 0, 0.0,
                                                        self.stream))
                    first = False
                weights = [w for unit in rnn._unit_params() for w in unit]
                p, g = _lib.LstmParams(), _lib.LstmGrads()
                p.num_layers, p.input_size, p.hidden_size = L, d_in, H
                self._deposit(named)
                named = []
                lg = [self._grad_like(w) for w in weights]
                for l in range(L):
                    p.w_ih[l], p.w_hh[l], p.b_ih[l], p.b_hh[l] = [weights[4 * l + k].data_ptr() for k in range(4)]
                    g.w_ih[l], g.w_hh[l], g.b_ih[l], g.b_hh[l] = [lg[4 * l + k].data_ptr() for k in range(4)]
                nbytes = lib.empose_lstm_train_workspace_bytes(C.byref(p), B, F)
                ws = self.ws(nbytes)
                _lib.check(lib.empose_lstm_train_bwd(C.byref(p), B, F, ctx['x0'].data_ptr(), d_in, ctx['lens32'].data_ptr(),
                                                     _ptr(ctx['c0']), ctx['lstm_save'].data_ptr(), dy.data_ptr(), None,
                                                     C.byref(g), ws.data_ptr(), nbytes, self.stream))
                named += list(zip(weights, lg))
            else:
                for v, sv, dpd, ld in zip(ctx['init_views'], ctx['init_saves'], (dpad, dspad), (68, 12)):
                    gi = [self._grad_like(p_) for p_ in v.parameter_list()]
                    self._mlp_bwd(v, ctx['x0'].data_ptr(), d_in, dpd.data_ptr(), ld, sv, gi, False, T)
                    named += list(zip(v.parameter_list(), gi))
            self._deposit(named)
            self._join(0)
            if 'bwd3' in self.side_parts:
                self._join(1)
        self.ctx = None
        total = loss_vals[4]
        keys = ('pose', 'shape', 'reconstruction', 'fk', 'total_loss')
        if as_tensors:
            vals = {k: loss_vals[j] for j, k in enumerate(keys)}
        else:
            host = loss_vals.tolist()
            vals = {k: host[j] for j, k in enumerate(keys)}
        return total, vals
