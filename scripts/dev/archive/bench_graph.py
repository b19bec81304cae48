"""Dev: the headline forward replayed as one HIP graph vs launched call by call."""
import sys, time
sys.path.insert(0, '.')
import torch
import bench
dev = torch.device('cuda:0')
net, model = bench.build_net(12, True, 4)
net = net.to(dev)
w, inputs = bench.make_inputs(net, dev, 1024, 32, seed=1000)
for _ in range(3):
    out = net.forward_tensors(*inputs)
torch.cuda.synchronize()
def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('eager ms/step', timeit(lambda: net.forward_tensors(*inputs)))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    net.forward_tensors(*inputs)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    gout = net.forward_tensors(*inputs)
torch.cuda.synchronize()
print('graph ms/step', timeit(g.replay))
print('max diff', float((gout['pose'] - out['pose']).abs().max()))
