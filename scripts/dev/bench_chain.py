"""Dev: time the SMPL evaluation kernels alone at T=32768 (fwd+bwd), optionally stopping the chain kernel early."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from em_pose_amd import _lib, synthetic
if os.environ.get('EMPOSE_LIB_PATH'):
    _lib.LIB_PATH = os.environ['EMPOSE_LIB_PATH']  # dev: a traced build of the library (scripts/dev/chain_trace.sh)
from em_pose_amd.bodymodels.smpl import SMPLLayer
from em_pose_amd.helpers.configuration import lgd_config
from em_pose_amd.nn.models import create_model
dev = torch.device('cuda:0')
model = synthetic.make_model()
net = create_model(lgd_config(12, False, 1, hidden=32), SMPLLayer(model)).to(dev).eval()
h = net._ensure_handle(dev)
lib = _lib.lib()
T, F = int(os.environ.get('T', 32768)), 32
g = torch.Generator(device='cpu').manual_seed(0)
th = (torch.randn(T, 66, generator=g) * 0.25).to(dev); be = torch.randn(T, 10, generator=g).to(dev)
o_r = torch.eye(3).expand(T // F, 12, 3, 3).contiguous().to(dev); o_t = (torch.randn(T // F, 12, 3, generator=g) * 0.02).to(dev)
tgt = torch.randn(T, 144, generator=g).to(dev); sc = torch.ones(T, device=dev)
pos, ori, jo = (torch.empty(T, n, device=dev) for n in (36, 108, 66))
gt, gb = torch.empty(T, 66, device=dev), torch.empty(T, 10, device=dev)
nb = lib.empose_smpl_workspace_bytes(h, T); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
def run():
    _lib.check(lib.empose_smpl_sensors_fwd_bwd(h, T, F, _lib.dptr(th), 66, _lib.dptr(be), 10, _lib.dptr(o_r), _lib.dptr(o_t),
               _lib.dptr(tgt), 144, _lib.dptr(sc), _lib.dptr(pos), _lib.dptr(ori), _lib.dptr(jo), _lib.dptr(gt), 66, _lib.dptr(gb), 10,
               _lib.dptr(ws), nb, None))
for _ in range(3): run()
torch.cuda.synchronize()
lib.empose_profile_enable(1)
for _ in range(10): run()
p = _lib.profile_read(); lib.empose_profile_enable(0)
print('stop', os.environ.get('EMPOSE_CHAIN_STOP', '0'), {k: round(v[0] / v[1] * 1000, 1) for k, v in p.items()}, 'us/launch')

if os.environ.get('EMPOSE_LIB_PATH') and hasattr(lib, 'empose_debug_chain_trace'):
    import ctypes as C
    tr = (C.c_longlong * 64)()
    fn = lib.empose_debug_chain_trace
    fn.argtypes = [C.POINTER(C.c_longlong)]
    assert fn(tr) == 0
    names = ['P0 stage', 'P2 chain', 'P3 skin', 'P4a normals', 'P4b sensors', 'P4c edges', 'P4d gather', 'P5 dv+parts',
             'P5c bones', 'P6 subtree', 'P7 dR/dJ']
    for b in range(2):
        t = [tr[b * 32 + i] for i in range(12)]
        print('block', 0 if b == 0 else 9000, 'total', t[11] - t[0], 'cycles:',
              ', '.join('%s %d' % (n, t[i + 1] - t[i]) for i, n in enumerate(names)))

if hasattr(lib, 'empose_debug_pair_trace'):
    import ctypes as C
    tr = (C.c_longlong * 64)()
    fn = lib.empose_debug_pair_trace
    fn.argtypes = [C.POINTER(C.c_longlong)]
    assert fn(tr) == 0
    names = ['P0 stage', 'P2 chain', 'P3 skin', 'P4a normals', 'P4b sensors', 'P4c edges', 'P4d gather', 'P5 dvp+chunks',
             'P5c bones', 'P6 subtree', 'P7 dR/dJ']
    for b in range(2):
        t = [tr[b * 32 + i] for i in range(12)]
        print('block', b, 'total', t[11] - t[0], 'cycles:', ', '.join('%s %d' % (n, t[i + 1] - t[i]) for i, n in enumerate(names)))

if os.environ.get('EMPOSE_LIB_PATH') and hasattr(lib, 'empose_debug_tile_trace'):
    import ctypes as C
    NWV = int(os.environ.get('TL_WAVES', 8))
    tr = (C.c_longlong * (NWV * 32))()
    fn = lib.empose_debug_tile_trace
    fn.argtypes = [C.POINTER(C.c_longlong)]
    assert fn(tr) == 0
    for w in range(NWV):
        t = [tr[w * 32 + i] for i in range(32)]
        t0 = tr[0]
        rounds = ' '.join('r%d[%d..%d]' % (r, t[4 + 2 * r] - t0, t[5 + 2 * r] - t0) for r in range(12) if t[5 + 2 * r] > 0)
        print('   sensor r0: loads %d skin %d frame %d bwd %d moments %d' % (t[14] - t[4], t[15] - t[14], t[16] - t[15], t[17] - t[16], t[5] - t[17]))
        print('wave', w, 'rod', t[1] - t0, 'bar', t[2] - t0, 'chain', t[3] - t0, rounds, 'sens_end', t[28] - t0, 'end', t[29] - t0)
