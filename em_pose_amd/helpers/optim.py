"""
Adam as one kernel launch per step (`empose_adam_step`): torch.optim.Adam semantics (amsgrad off, no weight decay),
what reference scripts/train.py:125-130 constructs.  torch's own Adam is a dozen multi-tensor kernels per step; on the
training path every other kernel is hand-written, so is this one.

* Parameters whose `.grad` is None are skipped, moments untouched, like torch does (the chunk tables are built over the
  parameters that have a gradient and rebuilt when that set changes).
* `param_groups[0]['lr']` is read at every step, so a schedule can be driven by assigning to it; `state_dict()` /
  `load_state_dict()` use torch.optim.Adam's layout (a checkpoint written by either restores into the other).
* The kernel updates parameters through raw device pointers: `tensor._version` does not move.  Everything that caches
  something derived from the weights (the folded inference handle of nn/models.py) is invalidated through
  `layers.BN_STATS_GENERATION`, bumped here.
"""
import torch

from em_pose_amd import _lib

_CHUNK = 4096   # ADAM_CHUNK in csrc/kernels.h


class HipAdam(object):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params or not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in self.params):
            raise _lib.EmposeError('HipAdam needs contiguous fp32 parameters on the GPU')
        self.param_groups = [{'params': self.params, 'lr': float(lr), 'betas': (float(betas[0]), float(betas[1])),
                              'eps': float(eps), 'weight_decay': 0, 'amsgrad': False}]
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.steps = 0
        self.dev = self.params[0].device
        self._active, self._tables = None, None
        n = len(self.params)
        # gradient pointer table: pinned staging + a persistent device table (no pageable copy on the hot path)
        self._g_host = torch.zeros(n, dtype=torch.int64).pin_memory()
        self._g = torch.zeros(n, dtype=torch.int64, device=self.dev)
        self._g_seen, self._g_copied = None, None

    # torch.optim.Optimizer vocabulary
    lr = property(lambda self: self.param_groups[0]['lr'],
                  lambda self, v: self.param_groups[0].__setitem__('lr', float(v)))
    betas = property(lambda self: self.param_groups[0]['betas'])
    eps = property(lambda self: self.param_groups[0]['eps'])

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _tables_for(self, active):
        if active != self._active:
            i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=self.dev)
            ps = [self.params[i] for i in active]
            ct, co = [], []
            for k, p in enumerate(ps):
                for off in range(0, p.numel(), _CHUNK):
                    ct.append(k)
                    co.append(off)
            self._tables = {'p': i64([p.data_ptr() for p in ps]), 'm': i64([self.exp_avg[i].data_ptr() for i in active]),
                            'v': i64([self.exp_avg_sq[i].data_ptr() for i in active]),
                            'sizes': i64([p.numel() for p in ps]),
                            'chunk_tensor': torch.tensor(ct, dtype=torch.int32, device=self.dev), 'chunk_offset': i64(co)}
            self._active, self._g_seen = active, None
        return self._tables

    def step(self):
        active = tuple(i for i, p in enumerate(self.params) if p.grad is not None)
        self.steps += 1
        _bump_weights_generation()
        if not active:
            return
        t = self._tables_for(active)
        ptrs = [self.params[i].grad.data_ptr() for i in active]
        if ptrs != self._g_seen:   # gradient tensors are re-allocated every eager step, static inside a HIP graph
            if self._g_copied is not None:
                self._g_copied.synchronize()   # the previous upload has left the staging buffer (host ran ahead)
            self._g_host[:len(ptrs)] = torch.tensor(ptrs, dtype=torch.int64)
            self._g[:len(ptrs)].copy_(self._g_host[:len(ptrs)], non_blocking=True)
            self._g_copied = torch.cuda.Event()
            self._g_copied.record()
            self._g_seen = ptrs
        g = self.param_groups[0]
        with torch.cuda.device(self.dev):
            _lib.check(_lib.lib().empose_adam_step(
                t['chunk_tensor'].numel(), t['p'].data_ptr(), self._g.data_ptr(), t['m'].data_ptr(), t['v'].data_ptr(),
                t['sizes'].data_ptr(), t['chunk_tensor'].data_ptr(), t['chunk_offset'].data_ptr(), g['lr'],
                g['betas'][0], g['betas'][1], g['eps'], self.steps, _lib.current_stream()))

    def state_dict(self):
        grp = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        grp['params'] = list(range(len(self.params)))
        return {'state': {i: {'step': torch.tensor(float(self.steps)), 'exp_avg': m, 'exp_avg_sq': v}
                          for i, (m, v) in enumerate(zip(self.exp_avg, self.exp_avg_sq))},
                'param_groups': [grp]}

    def load_state_dict(self, sd):
        """Restores moments, step count and hyper-parameters written by `state_dict()` or by torch.optim.Adam over the
        same parameter list.  The moment tensors are copied INTO the existing ones (their addresses are baked into the
        pointer tables and into captured graphs)."""
        grp = sd['param_groups'][0]
        if len(grp['params']) != len(self.params):
            raise ValueError('optimizer state for {} parameters, this optimizer has {}'.format(
                len(grp['params']), len(self.params)))
        state, steps = sd['state'], set()
        for k, idx in enumerate(grp['params']):
            st = state.get(idx, state.get(str(idx)))
            if st is None:          # torch keeps no state for a parameter that never had a gradient
                self.exp_avg[k].zero_()
                self.exp_avg_sq[k].zero_()
                continue
            if tuple(st['exp_avg'].shape) != tuple(self.params[k].shape):
                raise ValueError('moment shape mismatch for parameter {}'.format(k))
            self.exp_avg[k].copy_(st['exp_avg'])
            self.exp_avg_sq[k].copy_(st['exp_avg_sq'])
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError('per-parameter step counts differ ({}): one shared step count is supported'.format(
                sorted(steps)))
        self.steps = steps.pop() if steps else 0
        g = self.param_groups[0]
        g['lr'], g['eps'] = float(grp['lr']), float(grp['eps'])
        g['betas'] = (float(grp['betas'][0]), float(grp['betas'][1]))


def _bump_weights_generation():
    from em_pose_amd.nn import layers as _layers
    _layers.BN_STATS_GENERATION[0] += 1
