"""
Adam as one kernel launch per step (`empose_adam_step`): torch.optim.Adam semantics (amsgrad off, no weight decay),
what reference scripts/train.py:125-130 constructs.  torch's own Adam is a dozen multi-tensor kernels per step; on the
training path every other kernel is hand-written, so is this one.

* Parameters whose `.grad` is None are skipped, moments AND step count untouched, like torch does (torch keeps a step
  count per parameter: the bias correction of a parameter that sat out a step lags behind).  The chunk tables are
  built per group of parameters with the same step count -- one group, one launch, unless gradients were ever missing.
* `param_groups[0]['lr']` is read at every step, so a schedule can be driven by assigning to it; `state_dict()` /
  `load_state_dict()` use torch.optim.Adam's layout (a checkpoint written by either restores into a HipAdam).
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
        self.step_of = [0] * len(self.params)     # per parameter, as torch.optim.Adam keeps it
        self.dev = self.params[0].device
        self._groups = {}                         # tuple of parameter indices -> device tables
        self._stage = {}                          # tuple of parameter indices -> (pinned host table, event)

    # torch.optim.Optimizer vocabulary
    lr = property(lambda self: self.param_groups[0]['lr'],
                  lambda self, v: self.param_groups[0].__setitem__('lr', float(v)))
    betas = property(lambda self: self.param_groups[0]['betas'])
    eps = property(lambda self: self.param_groups[0]['eps'])

    @property
    def steps(self):
        """The step count when all parameters share one (the usual case)."""
        return max(self.step_of) if self.step_of else 0

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _tables_for(self, idx):
        t = self._groups.get(idx)
        if t is None:
            i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=self.dev)
            ps = [self.params[i] for i in idx]
            ct, co = [], []
            for k, p in enumerate(ps):
                for off in range(0, p.numel(), _CHUNK):
                    ct.append(k)
                    co.append(off)
            t = {'p': i64([p.data_ptr() for p in ps]), 'm': i64([self.exp_avg[i].data_ptr() for i in idx]),
                 'v': i64([self.exp_avg_sq[i].data_ptr() for i in idx]), 'sizes': i64([p.numel() for p in ps]),
                 'chunk_tensor': torch.tensor(ct, dtype=torch.int32, device=self.dev), 'chunk_offset': i64(co),
                 # gradient pointer table: pinned staging + a persistent device table (no pageable copy per step)
                 'g': torch.zeros(len(idx), dtype=torch.int64, device=self.dev),
                 'g_host': torch.zeros(len(idx), dtype=torch.int64).pin_memory(), 'g_seen': None, 'g_copied': None}
            self._groups[idx] = t
        return t

    def step(self):
        _bump_weights_generation()
        by_step = {}
        for i, p in enumerate(self.params):
            if p.grad is not None:
                by_step.setdefault(self.step_of[i] + 1, []).append(i)
        g = self.param_groups[0]
        for step, members in sorted(by_step.items()):
            idx = tuple(members)
            t = self._tables_for(idx)
            ptrs = [self.params[i].grad.data_ptr() for i in idx]
            if ptrs != t['g_seen']:   # gradient tensors are re-allocated every eager step, static inside a HIP graph
                if t['g_copied'] is not None:
                    t['g_copied'].synchronize()   # the previous upload has left the staging buffer (host ran ahead)
                t['g_host'].copy_(torch.tensor(ptrs, dtype=torch.int64))
                t['g'].copy_(t['g_host'], non_blocking=True)
                t['g_copied'] = torch.cuda.Event()
                t['g_copied'].record()
                t['g_seen'] = ptrs
            with torch.cuda.device(self.dev):
                _lib.check(_lib.lib().empose_adam_step(
                    t['chunk_tensor'].numel(), t['p'].data_ptr(), t['g'].data_ptr(), t['m'].data_ptr(),
                    t['v'].data_ptr(), t['sizes'].data_ptr(), t['chunk_tensor'].data_ptr(), t['chunk_offset'].data_ptr(),
                    g['lr'], g['betas'][0], g['betas'][1], g['eps'], step, _lib.current_stream()))
            for i in idx:
                self.step_of[i] = step

    def state_dict(self):
        grp = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        grp['params'] = list(range(len(self.params)))
        return {'state': {i: {'step': torch.tensor(float(self.step_of[i])), 'exp_avg': m, 'exp_avg_sq': v}
                          for i, (m, v) in enumerate(zip(self.exp_avg, self.exp_avg_sq)) if self.step_of[i] > 0},
                'param_groups': [grp]}

    def load_state_dict(self, sd):
        """Restores moments, step count and hyper-parameters written by `state_dict()` or by torch.optim.Adam over the
        same parameter list.  The moment tensors are copied INTO the existing ones (their addresses are baked into the
        pointer tables and into captured graphs)."""
        grp = sd['param_groups'][0]
        if len(grp['params']) != len(self.params):
            raise ValueError('optimizer state for {} parameters, this optimizer has {}'.format(
                len(grp['params']), len(self.params)))
        state = sd['state']
        for k, idx in enumerate(grp['params']):
            st = state.get(idx, state.get(str(idx)))
            if st is None:          # torch keeps no state for a parameter that never had a gradient
                self.exp_avg[k].zero_()
                self.exp_avg_sq[k].zero_()
                self.step_of[k] = 0
                continue
            if tuple(st['exp_avg'].shape) != tuple(self.params[k].shape):
                raise ValueError('moment shape mismatch for parameter {}'.format(k))
            self.exp_avg[k].copy_(st['exp_avg'])
            self.exp_avg_sq[k].copy_(st['exp_avg_sq'])
            self.step_of[k] = int(float(st['step']))
        g = self.param_groups[0]
        g['lr'], g['eps'] = float(grp['lr']), float(grp['eps'])
        g['betas'] = (float(grp['betas'][0]), float(grp['betas'][1]))


def _bump_weights_generation():
    from em_pose_amd.nn import layers as _layers
    _layers.BN_STATS_GENERATION[0] += 1
