"""Dev: does replaying the LGD-RNN-12 forward (B = 1024, F = 32) as a HIP graph beat eager launches?"""
import sys, time; sys.path.insert(0, '.')
import torch
import bench
dev = torch.device('cuda:0')
net, model = bench.build_net(12, True, 4)
net = net.to(dev)
w, inputs = bench.make_inputs(net, dev, 1024, 32, seed=1000)
for _ in range(3): out = net.forward_tensors(*inputs)
torch.cuda.synchronize()
def timed(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('eager: %.3f ms' % timed(lambda: net.forward_tensors(*inputs)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): net.forward_tensors(*inputs)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        out_g = net.forward_tensors(*inputs)
    print('graph: %.3f ms' % timed(g.replay))
    ref = net.forward_tensors(*inputs)
    g.replay(); torch.cuda.synchronize()
    print('max diff', max(float((ref[k] - out_g[k]).abs().max()) for k in ('pose', 'shape', 'joints')))
except Exception as e:
    print('capture failed:', type(e).__name__, str(e)[:300])
print('eager again: %.3f ms' % timed(lambda: net.forward_tensors(*inputs)))
