import sys; sys.path.insert(0, '.')
import torch, numpy as np
from em_pose_amd.helpers.optim import HipAdam
DEV='cuda:0'
torch.manual_seed(4)
shapes = [(300, 40), (66,), (5000,), (1,)]
ours = [torch.randn(*s, device=DEV).requires_grad_(True) for s in shapes]
ref = [p.detach().clone().requires_grad_(True) for p in ours]
oa, ob = HipAdam(ours, lr=5e-4), torch.optim.Adam(ref, lr=5e-4)
def one_step(oa, ob, ours, ref, skip):
    for i, (p, q) in enumerate(zip(ours, ref)):
        g = torch.randn_like(p)
        p.grad, q.grad = (None, None) if i == skip else (g.clone(), g.clone())
    oa.step(); ob.step()
one_step(oa, ob, ours, ref, None); one_step(oa, ob, ours, ref, 2); one_step(oa, ob, ours, ref, None)
torch.cuda.synchronize()
print('part1', [float((p-q).abs().max()) for p,q in zip(ours,ref)])
ours2 = [p.detach().clone().requires_grad_(True) for p in ours]
ref2 = [p.detach().clone().requires_grad_(True) for p in ours]
for p in ours + ref + ours2 + ref2: p.grad = None
oa2, ob2 = HipAdam(ours2, lr=1.0), torch.optim.Adam(ref2, lr=1.0)
a, b = HipAdam(ours, lr=5e-4), torch.optim.Adam(ref, lr=5e-4)
one_step(a, b, ours, ref, None); one_step(a, b, ours, ref, None)
print('after a,b 2 steps', [float((p-q).abs().max()) for p,q in zip(ours,ref)])
for p, q in zip(ours2, ours): p.data.copy_(q.data)
for p, q in zip(ref2, ref): p.data.copy_(q.data)
oa2.load_state_dict(b.state_dict()); ob2.load_state_dict(a.state_dict())
print(oa2.step_of, [float((m - b.state[q]['exp_avg']).abs().max()) for m, q in zip(oa2.exp_avg, ref)])
one_step(a, b, ours, ref, None)
gs = [p.grad.clone() for p in ours]
print('grad same', [float((p.grad-q.grad).abs().max()) for p,q in zip(ours,ref)])
for p, q, g in zip(ours2, ref2, gs): p.grad, q.grad = g.clone(), g.clone()
oa2.step(); ob2.step(); torch.cuda.synchronize()
print('ours2 vs ref', [float((p-q).abs().max()) for p,q in zip(ours2,ref)])
print('ref2 vs ours', [float((p-q).abs().max()) for p,q in zip(ref2,ours)])
print('ours vs ref', [float((p-q).abs().max()) for p,q in zip(ours,ref)])
