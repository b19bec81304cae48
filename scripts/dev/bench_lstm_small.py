"""Dev: the stand-alone LSTM entry point (2 x 512, input 60) on small batches: whole-sequence kernel vs step launches."""
import os, sys, time
sys.path.insert(0, '.')
import torch
from em_pose_amd import _lib
if os.environ.get('EMPOSE_LIB_PATH'):
    _lib.LIB_PATH = os.environ['EMPOSE_LIB_PATH']   # dev: a lab build of the library (scripts/dev/persist_lab.sh)
from em_pose_amd.nn.layers import RNNLayer
dev = 'cuda:0'
layer = RNNLayer(60, 512, 2).eval().to(dev)
for B in (1, 2, 3, 4, 6, 8, 12, 16):
    for F in (int(os.environ.get('F', 64)),):
        x = torch.randn(B, F, 60, device=dev)
        lens = torch.full((B,), F, device=dev)
        res = []
        for mode in ('1', '0'):
            _lib.check(_lib.lib().empose_set_option(b'lstm_persist', int(mode)))   # the library reads no environment
            for _ in range(3): layer(x, lens)
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(10): layer(x, lens)
            torch.cuda.synchronize(); res.append((time.time() - t0) / 10 * 1e3)
        print('B=%2d F=%3d: whole-sequence %.3f ms (%.2f us/step)   step launches %.3f ms (%.2f us/step)' % (
            B, F, res[0], res[0] * 1e3 / (F + 1), res[1], res[1] * 1e3 / (F + 1)))
_lib.check(_lib.lib().empose_set_option(b'lstm_persist', 1))
