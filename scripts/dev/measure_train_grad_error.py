"""Dev: how far are the training gradients from the reference's recorded ones, per tensor, in units of the tensor's
scale and of the reference's own one-ulp sensitivity?  (input for the tolerance of
tests/test_hip_parity.py::test_training_step_matches_reference_gradients)"""
import json, os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from tests import helpers as H
from tests.test_hip_parity import build_net, cfg_of, gpu, DEV
from em_pose_amd import _lib
from em_pose_amd.data.data import SyntheticBatch
sens_all = json.load(open(os.path.join(H.GOLDEN, 'train_sensitivity.json')))
for name in ('train_lgdrnn12_n2', 'train_lgd6_n2'):
    for fused in (0, 2):
        _lib.check(_lib.lib().empose_set_option(b'train_fused', fused))
        case = H.load_case(name); meta, w, rec = case['meta'], case['in'], case['run']
        net = build_net(cfg_of(meta), H.small_model(), meta['vertex_ids'], case['sd']); net.train()
        batch = SyntheticBatch(w, torch.from_numpy(w['seq_lengths']).to(DEV), device=DEV); batch.joints_gt = gpu(w['joints_gt'])
        net.zero_grad(); out = net(batch); net.backward(batch, out)
        sens = sens_all[name]
        gmax = max(np.abs(v).max() for kk, v in rec.items() if kk.startswith('grad/'))
        worst_scale, worst_sens, worst_k = 0.0, 0.0, None
        rows = []
        for k, p in net.named_parameters():
            want = rec.get('grad/' + k)
            if k.startswith('smpl.') or want is None: continue
            if k.endswith('.bias') and ('input_to_hidden' in k or '.layers.0.' in k or '.layers.4.' in k): continue
            got = p.grad.detach().cpu().numpy()
            err = np.abs(got - want).max(); scale = max(np.abs(want).max(), 1e-4 * gmax); s = sens['grad'].get(k, 0.0)
            rows.append((err / scale, err / s if s > 0 else float('inf'), k))
        rows.sort(reverse=True)
        print(name, 'fused' if fused else 'plain', 'worst err/scale %.2e' % rows[0][0],
              '| worst err/sens among the top: %.2f' % max(r[1] for r in rows[:5]))
        for r in rows[:4]: print('    err/scale %.2e  err/sens %.2f  %s' % r)
        # the criterion that would pass: err <= max(a * scale, b * sens): smallest a with b = 4
        need_a = max((r[0] for r in rows if r[1] > 4.0), default=0.0)
        print('    smallest a with tol = max(a * scale, 4 * sens): %.2e' % need_a)
_lib.lib().empose_set_option(b'train_fused', 0)
