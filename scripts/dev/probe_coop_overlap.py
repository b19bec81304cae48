"""Dev probe: does the whole-sequence (cooperative) LSTM launch of a B = 1, F = 256 forward run beside kernels of another
stream?  (a) two LGD-RNN forwards on one stream vs two streams; (b) an LGD-RNN forward + an LGD (MLP-init) forward."""
import sys, time; sys.path.insert(0, '.')
import ctypes as C, torch
import bench
from em_pose_amd import _lib
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
lib = _lib.lib()
B, F = 1, 256
def make_call(net, stream, seed):
    h = net._ensure_handle(dev)
    w, inputs = bench.make_inputs(net, dev, B, F, seed=seed)
    mp, mo, ot, orr = [t.contiguous() for t in inputs]
    out = [torch.empty(B, F, n, device=dev) for n in (66, 10, 66)]
    hn = [torch.empty(2, B, 512, device=dev) for _ in range(2)]
    io = _lib.LgdIO(); io.B, io.F = B, F
    io.marker_pos, io.marker_oris, io.offset_t, io.offset_r = [_lib.dptr(t) for t in (mp, mo, ot, orr)]
    io.pose_hat, io.shape_hat, io.joints_hat = [_lib.dptr(t) for t in out]
    if net.rnn_init: io.h_n, io.c_n = _lib.dptr(hn[0]), _lib.dptr(hn[1])
    nb = lib.empose_lgd_workspace_bytes(h, B, F); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    keep = (mp, mo, ot, orr, out, hn, ws, io)
    def call():
        _lib.check(lib.empose_lgd_forward(h, C.byref(io), _lib.dptr(ws), nb, C.c_void_p(stream.cuda_stream)))
    return call, keep
def timeit(calls, n=20):
    for c in calls: c()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for c in calls: c()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
rnn, _ = bench.build_net(12, True, 4); rnn = rnn.to(dev)
mlp, _ = bench.build_net(12, False, 4); mlp = mlp.to(dev)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
a0, k0 = make_call(rnn, s0, 1); a1, k1 = make_call(rnn, s1, 2); a0b, k2 = make_call(rnn, s0, 3)
m0, k3 = make_call(mlp, s0, 4); m1, k4 = make_call(mlp, s1, 5)
print('LGD-RNN forward alone                 : %.3f ms' % timeit([a0]))
print('LGD (MLP init) forward alone          : %.3f ms' % timeit([m0]))
print('two LGD-RNN forwards, one stream      : %.3f ms' % timeit([a0, a0b]))
print('two LGD-RNN forwards, two streams     : %.3f ms' % timeit([a0, a1]))
print('LGD-RNN + LGD forwards, one stream    : %.3f ms' % timeit([a0, m0]))
print('LGD-RNN + LGD forwards, two streams   : %.3f ms' % timeit([a0, m1]))
