"""Dev (CPU): accuracy of a dot product computed from three-way bf16 splits of its fp32 operands (six piece products,
fp32 accumulation), against plain fp32 accumulation, both measured against fp64.  See DESIGN.md 9.8."""
import numpy as np
rng = np.random.default_rng(0)

def bf16(x):   # round-to-nearest-even to bfloat16, returned as float32
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)

def split3(x):
    h = bf16(x); r = x - h; m = bf16(r); l = bf16(r - m)
    return h, m, l

M, N, K = 256, 512, 512
A = rng.normal(size=(M, K)).astype(np.float32)
W = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
ref = A.astype(np.float64) @ W.astype(np.float64).T
f32 = np.zeros((M, N), np.float32)
for k0 in range(0, K, 8):   # fp32 accumulation in k chunks, like an MFMA chain
    f32 += A[:, k0:k0 + 8] @ W[:, k0:k0 + 8].T
a, w = split3(A), split3(W)
terms = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
for name, sel in (('bf16x3, 6 products', terms), ('bf16x3, 3 products', terms[:3]), ('bf16 x bf16', terms[:1])):
    acc = np.zeros((M, N), np.float32)
    for k0 in range(0, K, 16):
        for i, j in sel[::-1]:   # small terms first
            acc += a[i][:, k0:k0 + 16] @ w[j][:, k0:k0 + 16].T
    print('%-20s max abs err %.3e   (fp32 accumulation: %.3e, |ref| max %.2f)' % (
        name, np.abs(acc - ref).max(), np.abs(f32 - ref).max(), np.abs(ref).max()))
