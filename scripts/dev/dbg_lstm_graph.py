"""Dev: does a forward+backward of nn.LSTM(144, 512, 2) on (T, B, 144) capture into a HIP graph?  usage: T B  [NO_MIOPEN=1]
(MIOpen path: T <= 31 captures, T >= 32 crashes in hipStreamEndCapture; the native path captures.)"""
import sys, torch
T, B = int(sys.argv[1]), int(sys.argv[2])
import os
if os.environ.get("NO_MIOPEN"): torch.backends.cudnn.enabled = False
lstm = torch.nn.LSTM(144, 512, 2).cuda()
x = torch.randn(T, B, 144, device='cuda')
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        lstm.zero_grad(set_to_none=True); y, _ = lstm(x); y.sum().backward()
    s.synchronize(); lstm.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    y, _ = lstm(x); loss = y.sum(); loss.backward()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print('ok', T, B, float(loss), flush=True)
