"""Dev: the one LGD fuzz case above 1e-5 (seed 3602, case 20: B=257, F=3, masks, carried state) stage by stage --
HIP vs the float64 oracle next to the fp32 oracle vs the float64 oracle, per iteration and per quantity."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import helpers as H
from em_pose_amd import _lib, synthetic
from oracle import torch_ref as R
from tests.fuzz import fuzz_lgd

seed, case_no = int(sys.argv[1]) if len(sys.argv) > 1 else 3602, int(sys.argv[2]) if len(sys.argv) > 2 else 20
tile, fuse = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 0)
dev = 'cuda:0'
rng = np.random.default_rng(seed); torch.manual_seed(seed)
model, bm, nets = fuzz_lgd.build_nets(dev, False)
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
print(name, dict(B=B, F=F, masks=masks is not None, state=state is not None), 'lens', np.bincount(lens))
def sensors(poses, betas, o_r, o_t):
    with torch.no_grad():
        p, o, _ = R.estimated_markers(bm, tables, vids, torch.from_numpy(poses), torch.from_numpy(betas), torch.from_numpy(o_r), torch.from_numpy(o_t))
    return p.numpy(), o.numpy()
w = synthetic.make_windows(B, F, 1000 + case_no, sensors)
if masks is not None: w['marker_masks'] = masks
kw = dict(n_markers=int(meta['n_markers']), N=int(meta['N']), rnn_init=rnn)
inp = H.oracle_inputs(w, sl=lens)
_, h32 = R.ief_forward(sd, bm, tables, vids, inp, rnn_state=state, **kw)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
_, h64 = R.ief_forward(sd64, R.BodyModelTensors(model, dtype=torch.float64), tables, vids, H.oracle_inputs(w, sl=lens, dtype=torch.float64),
                       rnn_state=None if state is None else tuple(t.double() for t in state), **kw)
lib = _lib.lib()
_lib.check(lib.empose_set_option(b'smpl_tile', tile)); _lib.check(lib.empose_set_option(b'smpl_fuse', fuse))
g = lambda t: None if t is None else t.to(dev)
res = net.forward_tensors(g(inp['marker_pos']), g(inp['marker_oris']), g(inp['offset_t']), g(inp['offset_r']), marker_masks=g(inp['marker_masks']),
                          seq_lengths=g(inp['seq_lengths']), state=None if state is None else tuple(g(t) for t in state), keep_history=True, keep_gradient_trace=True)
torch.cuda.synchronize()
valid = (torch.arange(F)[None, :] < torch.as_tensor(lens)[:, None]).reshape(-1).numpy()
fm = valid.copy()
if masks is not None:
    fm &= (masks.reshape(B * F, 12) == 1).all(axis=1)
print('valid frames', valid.sum(), 'of', B * F, '; frames with every sensor present', fm.sum())
N = kw['N']
def err(a, b, m): 
    d = np.abs(np.asarray(a, dtype=np.float64).reshape(B * F, -1) - np.asarray(b, dtype=np.float64).reshape(B * F, -1))[m]
    return d.max() if d.size else 0.0
print('%-14s %-4s %12s %12s %12s' % ('quantity', 'it', 'hip-f64', 'o32-f64', 'scale'))
for i in range(N + 1):
    for key, hk in (('pose', 'pose'), ('shape', 'shape'), ('joints', 'joints'), ('markers', 'markers'), ('markers_ori', 'markers_ori')):
        a = res['hist'][hk][i].cpu().numpy(); r64 = h64[key][i].numpy(); r32 = h32[key][i].numpy()
        print('%-14s %-4d %12.3e %12.3e %12.3e' % (key, i, err(a, r64, valid), err(r32, r64, valid), np.abs(r64.reshape(B * F, -1)[valid]).max()))
    if i < N:
        for key, hk in (('g_pose', 'g_pose'), ('g_shape', 'g_shape')):
            a = res['trace'][hk][i].cpu().numpy(); r64 = h64[key][i].numpy(); r32 = h32[key][i].numpy()
            d = np.abs(a.astype(np.float64).reshape(B * F, -1) - r64.reshape(B * F, -1))
            worst = np.unravel_index(np.argmax(d * valid[:, None]), d.shape)
            print('%-14s %-4d %12.3e %12.3e %12.3e   worst at frame %d (window %d, len %d, full sensors %s) col %d: hip %.6e o32 %.6e f64 %.6e'
                  % (key, i, err(a, r64, valid), err(r32, r64, valid), np.abs(r64.reshape(B * F, -1)[valid]).max(), worst[0], worst[0] // F, lens[worst[0] // F], fm[worst[0]], worst[1],
                     a.reshape(B * F, -1)[worst], r32.reshape(B * F, -1)[worst], r64.reshape(B * F, -1)[worst]))
if rnn:
    print('lstm state h: hip-f64 %.3e  o32-f64 %.3e' % (float((res['state'][0].cpu().double() - h64['rnn_state'][0]).abs().max()), float((h32['rnn_state'][0].double() - h64['rnn_state'][0]).abs().max())))
print('per-frame relative error of g_pose (max over the 66 columns / max |g| of the frame), frames with every sensor present:')
for i in range(N):
    a = res['trace']['g_pose'][i].cpu().numpy().astype(np.float64).reshape(B * F, -1); r64 = h64['g_pose'][i].numpy().reshape(B * F, -1); r32 = h32['g_pose'][i].numpy().astype(np.float64).reshape(B * F, -1)
    m = fm & (np.abs(r64).max(axis=1) > 0)
    sc = np.abs(r64[m]).max(axis=1)
    eh, eo = np.abs(a[m] - r64[m]).max(axis=1) / sc, np.abs(r32[m] - r64[m]).max(axis=1) / sc
    q = lambda v: ' '.join('%.2e' % np.quantile(v, p) for p in (0.5, 0.9, 0.99, 1.0))
    print('  it %d  hip: %s | o32: %s | frames where hip is worse: %d of %d; median ratio hip/o32 %.2f' % (i, q(eh), q(eo), int((eh > eo).sum()), m.sum(), np.median(eh / np.maximum(eo, 1e-12))))
# forward pieces of frame 192 at iteration 0: orientation error per sensor
a = res['hist']['markers_ori'][0].cpu().numpy().astype(np.float64).reshape(B * F, 12, 9); r64 = h64['markers_ori'][0].numpy().reshape(B * F, 12, 9); r32 = h32['markers_ori'][0].numpy().astype(np.float64).reshape(B * F, 12, 9)
wf = 192 if B * F > 192 else 0
print('frame %d orientation error per sensor, hip:' % wf, np.abs(a[wf] - r64[wf]).max(axis=1).round(8))
print('frame %d orientation error per sensor, o32:' % wf, np.abs(r32[wf] - r64[wf]).max(axis=1).round(8))
