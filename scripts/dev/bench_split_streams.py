"""Dev experiment: one batch of 1024 windows vs two halves on two streams (windows are independent)."""
import sys, time; sys.path.insert(0, '.')
import ctypes as C, torch
import bench
from em_pose_amd import _lib
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
net, model = bench.build_net(12, True, 4); net = net.to(dev)
B, F = 1024, 32
w, inputs = bench.make_inputs(net, dev, B, F, seed=1000)
lib = _lib.lib(); h = net._ensure_handle(dev)
def make_call(sl, stream):
    mp, mo, ot, orr = [t[sl].contiguous() for t in inputs]
    b = mp.shape[0]
    out = [torch.empty(b, F, n, device=dev) for n in (66, 10, 66)]
    hn = [torch.empty(2, b, 512, device=dev) for _ in range(2)]
    io = _lib.LgdIO(); io.B, io.F = b, F
    io.marker_pos, io.marker_oris, io.offset_t, io.offset_r = [_lib.dptr(t) for t in (mp, mo, ot, orr)]
    io.pose_hat, io.shape_hat, io.joints_hat = [_lib.dptr(t) for t in out]
    io.h_n, io.c_n = _lib.dptr(hn[0]), _lib.dptr(hn[1])
    nb = lib.empose_lgd_workspace_bytes(h, b, F); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    keep = (mp, mo, ot, orr, out, hn, ws, io)
    def call():
        _lib.check(lib.empose_lgd_forward(h, C.byref(io), _lib.dptr(ws), nb, C.c_void_p(stream.cuda_stream)))
    return call, keep
def timeit(calls, n=10):
    for c in calls: c()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for c in calls: c()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
s0 = torch.cuda.current_stream()
full, k0 = make_call(slice(0, B), s0)
print('full batch, 1 stream      : %.3f ms' % timeit([full]))
for parts in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    calls = [make_call(slice(i * B // parts, (i + 1) * B // parts), streams[i]) for i in range(parts)]
    print('%d parts on %d streams      : %.3f ms' % (parts, parts, timeit([c for c, _ in calls])))
    calls1 = [make_call(slice(i * B // parts, (i + 1) * B // parts), s0) for i in range(parts)]
    print('%d parts on 1 stream       : %.3f ms' % (parts, timeit([c for c, _ in calls1])))
