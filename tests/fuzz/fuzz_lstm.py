"""Dev: random shapes through empose_rnn_fwd against torch.nn.LSTM on the CPU (packed sequences): all three batch
regimes (whole-sequence, K-split, chain kernels), uni/bi-directional, ragged lengths, given state."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
from em_pose_amd.nn.layers import RNNLayer
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
t_end, n, worst = time.time() + budget, 0, 0.0
while time.time() < t_end:
    bi = bool(rng.integers(0, 2)); L = int(rng.integers(1, 5 if not bi else 3))
    H = int(rng.integers(1, 17)) * 4; In = int(rng.integers(1, 40)) * 4
    B = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 31, 33, 64, 100, 256, 257, 300, 700])); F = int(rng.integers(1, 24))
    # round 3: above 256 rows the whole-sequence cooperative kernel (opt-in) in half of the cases, small shapes included
    from em_pose_amd import _lib
    _lib.check(_lib.lib().empose_set_option(b'lstm_seq', int(rng.integers(0, 2))))
    torch.manual_seed(n)
    layer = RNNLayer(In, H, L, bidirectional=bi).eval()
    with torch.no_grad():
        for p in layer.lstm.parameters(): p.mul_(2.0)
    x = torch.randn(B, F, In); lens = torch.randint(1, F + 1, (B,)); lens[0] = F
    U = L * (2 if bi else 1)
    state = None if rng.integers(0, 2) else (0.5 * torch.randn(U, B, H), 0.5 * torch.randn(U, B, H))
    with torch.no_grad():
        ref, (rh, rc) = layer.lstm(pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False), state)
        ref, _ = pad_packed_sequence(ref, batch_first=True, total_length=F)
    g = layer.to('cuda:0'); g.init_state = None if state is None else tuple(t.cuda() for t in state)
    got = g(x.cuda(), lens.cuda()); torch.cuda.synchronize()
    err = max(float((got.cpu() - ref).abs().max()), float((g.final_state[0].cpu() - rh).abs().max()),
              float((g.final_state[1].cpu() - rc).abs().max()))
    worst = max(worst, err); n += 1
    if not err < 1e-4:
        print('MISMATCH', dict(bi=bi, L=L, H=H, In=In, B=B, F=F, state=state is not None), err); sys.exit(1)
    g.release()
print('%d random cases, worst abs error %.2e' % (n, worst))
