"""
Adam as one kernel launch per step (`empose_adam_step`): torch.optim.Adam semantics (amsgrad off, no weight decay),
what reference scripts/train.py:125-130 constructs.  torch's own Adam is a dozen multi-tensor kernels per step; on the
training path every other kernel is hand-written, so is this one.  Parameters without a gradient are skipped, like
torch does.
"""
import torch

from em_pose_amd import _lib

_CHUNK = 4096   # ADAM_CHUNK in csrc/kernels.h


class HipAdam(object):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params or not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in self.params):
            raise _lib.EmposeError('HipAdam needs contiguous fp32 parameters on the GPU')
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.steps = 0
        dev = self.params[0].device
        self.dev = dev
        i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)
        self._p = i64([p.data_ptr() for p in self.params])
        self._m = i64([t.data_ptr() for t in self.exp_avg])
        self._v = i64([t.data_ptr() for t in self.exp_avg_sq])
        self._sizes = i64([p.numel() for p in self.params])
        ct, co = [], []
        for i, p in enumerate(self.params):
            for off in range(0, p.numel(), _CHUNK):
                ct.append(i)
                co.append(off)
        self._chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=dev)
        self._chunk_offset = i64(co)
        self._g, self._g_host = None, None

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self):
        if any(p.grad is None for p in self.params):
            raise _lib.EmposeError('HipAdam.step(): a parameter has no gradient')
        ptrs = [p.grad.data_ptr() for p in self.params]
        if ptrs != self._g_host:   # gradient tensors are re-allocated every eager step, static inside a HIP graph
            self._g = torch.tensor(ptrs, dtype=torch.int64, device=self.dev)
            self._g_host = ptrs
        self.steps += 1
        with torch.cuda.device(self.dev):
            _lib.check(_lib.lib().empose_adam_step(
                self._chunk_tensor.numel(), self._p.data_ptr(), self._g.data_ptr(), self._m.data_ptr(),
                self._v.data_ptr(), self._sizes.data_ptr(), self._chunk_tensor.data_ptr(), self._chunk_offset.data_ptr(),
                self.lr, self.betas[0], self.betas[1], self.eps, self.steps, _lib.current_stream()))

    def state_dict(self):
        return {'state': {i: {'step': self.steps, 'exp_avg': m, 'exp_avg_sq': v}
                          for i, (m, v) in enumerate(zip(self.exp_avg, self.exp_avg_sq))},
                'param_groups': [{'lr': self.lr, 'betas': self.betas, 'eps': self.eps,
                                  'params': list(range(len(self.params)))}]}
