import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import helpers as H
from em_pose_amd import synthetic
from oracle import torch_ref as R
from tests.fuzz import fuzz_lgd
seed, case_no = 3602, 20
rng = np.random.default_rng(seed); torch.manual_seed(seed)
model, bm, nets = fuzz_lgd.build_nets('cpu', False)
print({k: (v.shape, v.dtype) for k, v in model.items()})
for n in range(case_no + 1):
    name = list(nets)[int(rng.integers(0, len(nets)))]
    net, sd, meta, vids, tables = nets[name]
    B = int(rng.choice(list(fuzz_lgd.BATCHES))); F = int(rng.integers(1, 20))
    lens = rng.integers(1, F + 1, size=B); lens[0] = F
    masks = (rng.uniform(size=(B, F, 12)) > 0.05).astype(np.float32) if rng.integers(0, 2) else None
    rnn = bool(meta['rnn']); state = None
    if rnn and rng.integers(0, 2):
        state = (0.3 * torch.randn(2, B, 32), 0.3 * torch.randn(2, B, 32))
    rng.choice([0, 2]); rng.integers(0, 2)
    if masks is not None: rng.integers(0, 2)
    rng.integers(0, 2)
def sensors(poses, betas, o_r, o_t):
    with torch.no_grad():
        p, o, _ = R.estimated_markers(bm, tables, vids, torch.from_numpy(poses), torch.from_numpy(betas), torch.from_numpy(o_r), torch.from_numpy(o_t))
    return p.numpy(), o.numpy()
w = synthetic.make_windows(B, F, 1000 + case_no, sensors)
w['marker_masks'] = masks
kw = dict(n_markers=int(meta['n_markers']), N=int(meta['N']), rnn_init=rnn)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
st64 = tuple(t.double() for t in state)
valid = (torch.arange(F)[None, :] < torch.as_tensor(lens)[:, None]).numpy()
inp64 = H.oracle_inputs(w, sl=lens, dtype=torch.float64)
want64, _ = R.ief_forward(sd64, R.BodyModelTensors(model, dtype=torch.float64), tables, vids, inp64, rnn_state=st64, **kw)
prng = np.random.default_rng(1)
for what in ('model', 'inputs', 'weights'):
    sens = 0.0
    for draw in range(4):
        m2, w2, sd2 = model, w, sd64
        if what == 'model':
            m2 = {k: (v.astype(np.float64) * (1 + 6e-8 * np.sign(prng.standard_normal(v.shape))) if v.dtype.kind == 'f' else v) for k, v in model.items()}
        if what == 'inputs':
            w2 = dict(w)
            for k in ('marker_pos', 'marker_oris', 'offset_t'):
                w2[k] = w[k].astype(np.float64) * (1 + 6e-8 * np.sign(prng.standard_normal(w[k].shape)))
        if what == 'weights':
            sd2 = {k: (v * (1 + 6e-8 * torch.sign(torch.randn(v.shape, dtype=torch.float64))) if v.is_floating_point() else v) for k, v in sd64.items()}
        o, _ = R.ief_forward(sd2, R.BodyModelTensors(m2, dtype=torch.float64), tables, vids, H.oracle_inputs(w2, sl=lens, dtype=torch.float64), rnn_state=st64, **kw)
        s = max(float(np.abs(o[k].numpy() - want64[k].numpy())[valid].max()) for k in want64)
        sens = max(sens, s)
    print(what, 'one-ulp sensitivity', sens)
