"""Dev: time the LSTM alone (B=1024, F=32, 2x512) through empose_lstm_fwd."""
import os, sys; sys.path.insert(0, '.')
import torch
from em_pose_amd import _lib, synthetic
if os.environ.get('EMPOSE_LIB_PATH'):
    _lib.LIB_PATH = os.environ['EMPOSE_LIB_PATH']   # dev: a lab build of the library
from em_pose_amd.bodymodels.smpl import SMPLLayer
from em_pose_amd.helpers.configuration import lgd_config
from em_pose_amd.nn.models import create_model
dev = torch.device('cuda:0')
net = create_model(lgd_config(12, True, 4), SMPLLayer(synthetic.make_model(nu=8, nv=20, seed=160))).to(dev).eval()
net.vertex_ids = synthetic.small_vertex_ids(160)
h = net._ensure_handle(dev); lib = _lib.lib()
if len(sys.argv) > 1:
    _lib.check(lib.empose_set_option(b'lstm_seq', int(sys.argv[1])))
if os.environ.get('EMPOSE_LSTM_X3'):
    _lib.check(lib.empose_set_option(b'lstm_x3', int(os.environ['EMPOSE_LSTM_X3'])))
B, F = 1024, 32
x = torch.randn(B, F, 144, device=dev); y = torch.empty(B, F, 512, device=dev)
nb = lib.empose_lstm_workspace_bytes(h, B, F); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
def run(): _lib.check(lib.empose_lstm_fwd(h, B, F, _lib.dptr(x), 144, None, None, None, _lib.dptr(y), None, None, _lib.dptr(ws), nb, None))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
print('LSTM B=%d F=%d: %.3f ms  (%.1f us per wavefront launch, %.1f TFLOP/s)' % (B, F, t, t * 1e3 / 33, 2.0 * B * F * 2048 * (144 + 512 + 1024) / t / 1e9))
