"""Dev: empose_update_nets_fwd alone (both update MLPs, T rows), back-to-back launches, real model weights of the bench."""
import os, sys
sys.path.insert(0, '.')
import torch
from em_pose_amd import _lib, synthetic
from em_pose_amd.bodymodels.smpl import SMPLLayer
from em_pose_amd.helpers.configuration import lgd_config
from em_pose_amd.nn.models import create_model
dev = torch.device('cuda:0')
T = int(os.environ.get('T', 32768))
torch.manual_seed(0)
net = create_model(lgd_config(12, True, 4), SMPLLayer(synthetic.make_model())).to(dev).eval()
h = net._ensure_handle(dev)
lib = _lib.lib()
flush = torch.empty(160 * 1024 * 1024, device=dev)  # 640 MB: evicts L2 and the 256 MB Infinity Cache
for kind in ('randn', 'zeros', 'randn+flush'):
    x = torch.zeros(T, 296, device=dev) if kind == 'zeros' else torch.randn(T, 296, device=dev)
    dp, ds = torch.empty(T, 66, device=dev), torch.empty(T, 10, device=dev)
    nb = lib.empose_update_workspace_bytes(h, T); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    def run():
        _lib.check(lib.empose_update_nets_fwd(h, T, _lib.dptr(x), 296, _lib.dptr(dp), _lib.dptr(ds), _lib.dptr(ws), nb, None))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        if kind.endswith('flush'):
            tot = 0.0
            for _ in range(10):
                flush.fill_(1.0)
                e0.record(); run(); e1.record(); torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            ts.append(tot / 10)
            continue
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    flops = 2.0 * T * ((296 * 512 + 4 * 512 * 512 + 512 * 66) + (296 * 512 + 4 * 512 * 512 + 512 * 10))
    t = min(ts)
    print('x=%s T=%d: %.1f us/launch  %.1f TFLOP/s' % (kind, T, t * 1e3, flops / t / 1e9))

# sustained load: does the per-launch time drift once the chip is warm?
x = torch.randn(T, 296, device=dev)
for block in range(6):
    e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize()
    print('sustained block %d: %.1f us/launch' % (block, e0.elapsed_time(e1) / 100 * 1e3))

# what precedes the launch: the same call timed alone after ~1 ms of other work (the in-step situation)
big = torch.randn(64 * 1024 * 1024, device=dev)
for name, other in (('idle 2 ms', lambda: torch.cuda._sleep(int(2e-3 * 2.4e9))),
                    ('elementwise over 256 MB x4', lambda: [big.mul_(1.0001) for _ in range(4)]),
                    ('nothing', lambda: None)):
    tot = 0.0
    for _ in range(20):
        other()
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    print('after %-28s: %.1f us/launch' % (name, tot / 20 * 1e3))
