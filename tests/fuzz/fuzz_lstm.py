"""Random shapes through empose_rnn_fwd against torch.nn.LSTM on the CPU (packed sequences): all three batch regimes
(whole-sequence, K-split, chain kernels), uni/bi-directional, ragged lengths, given state.

    python tests/fuzz/fuzz_lstm.py <seed> <seconds | n=CASES>

`run()` is shared with tests/test_fuzz_slice.py (a fixed-seed slice inside `pytest -m gpu`)."""
import sys
import time

if __name__ == '__main__':
    sys.path.insert(0, '.')

import numpy as np
import torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

BATCHES = (1, 2, 3, 4, 5, 8, 12, 16, 17, 31, 33, 36, 64, 100, 128, 256, 257, 300, 700)   # (9 .. 64: lstm_mid16_x3.hip, .. 256: lstm_mid_x3.hip since round 6)
BATCHES_R5 = (1, 2, 3, 4, 5, 8, 12, 16, 17, 31, 33, 64, 100, 256, 257, 300, 700)   # (4, 12: the other two sizes of lstm_fewrows_kernel)


def run(seed=0, seconds=None, n_cases=None, batches=BATCHES, log=print, tol=1e-4):
    from em_pose_amd import _lib
    from em_pose_amd.nn.layers import RNNLayer
    rng = np.random.default_rng(seed)
    opt_rng = np.random.default_rng(seed + 7919)     # (its own stream: the cases of a seed stay what they were)
    t_end = time.time() + (seconds if seconds is not None else 1e9)
    n, worst, worst_case, above = 0, 0.0, None, []
    lib = _lib.lib()
    try:
        while time.time() < t_end and (n_cases is None or n < n_cases):
            bi = bool(rng.integers(0, 2))
            L = int(rng.integers(1, 5 if not bi else 3))
            H = int(rng.integers(1, 17)) * 4
            In = int(rng.integers(1, 40)) * 4
            B = int(rng.choice(list(batches)))
            F = int(rng.integers(1, 24))
            # above 256 rows the whole-sequence cooperative kernel (opt-in) in half of the cases, small shapes included
            seq = int(rng.integers(0, 2))
            _lib.check(lib.empose_set_option(b'lstm_seq', seq))
            # round 6: 9 .. 64 rows on 16-column tiles (default) or 32-column ones; 4 .. 64 rows as one cooperative launch (opt-in)
            tiles16, one_launch = int(opt_rng.integers(0, 2)), int(opt_rng.integers(0, 2))
            _lib.check(lib.empose_set_option(b'lstm_mid16', tiles16))
            _lib.check(lib.empose_set_option(b'lstm_midseq', one_launch))
            torch.manual_seed(n)
            layer = RNNLayer(In, H, L, bidirectional=bi).eval()
            with torch.no_grad():
                for p in layer.lstm.parameters():
                    p.mul_(2.0)
            x = torch.randn(B, F, In)
            lens = torch.randint(1, F + 1, (B,))
            lens[0] = F
            U = L * (2 if bi else 1)
            state = None if rng.integers(0, 2) else (0.5 * torch.randn(U, B, H), 0.5 * torch.randn(U, B, H))
            with torch.no_grad():
                ref, (rh, rc) = layer.lstm(pack_padded_sequence(x, lens, batch_first=True, enforce_sorted=False), state)
                ref, _ = pad_packed_sequence(ref, batch_first=True, total_length=F)
            g = layer.to('cuda:0')
            g.init_state = None if state is None else tuple(t.cuda() for t in state)
            got = g(x.cuda(), lens.cuda())
            torch.cuda.synchronize()
            err = max(float((got.cpu() - ref).abs().max()), float((g.final_state[0].cpu() - rh).abs().max()),
                      float((g.final_state[1].cpu() - rc).abs().max()))
            desc = dict(bi=bi, L=L, H=H, In=In, B=B, F=F, state=state is not None, lstm_seq=seq, lstm_mid16=tiles16,
                        lstm_midseq=one_launch)
            if err > 1e-5 or not np.isfinite(err):
                above.append((n, err, desc))
                log('lstm case %d above 1e-5: %.3e %s' % (n, err, desc))
            if err > worst:
                worst_case = (n, desc)
            worst = max(worst, err) if np.isfinite(err) else float('nan')
            n += 1
            g.release()
            assert err < tol, 'LSTM MISMATCH seed %d case %d %s: %r' % (seed, n - 1, desc, err)
    finally:
        _lib.check(lib.empose_set_option(b'lstm_seq', 0))
        _lib.check(lib.empose_set_option(b'lstm_mid16', 1))
        _lib.check(lib.empose_set_option(b'lstm_midseq', 0))
    return {'n': n, 'worst': worst, 'worst_case': worst_case, 'above_1e5': above}


if __name__ == '__main__':
    arg = sys.argv[2] if len(sys.argv) > 2 else '60'
    r = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, n_cases=int(arg[2:]) if arg.startswith('n=') else None,
            seconds=None if arg.startswith('n=') else float(arg))
    print('%d random cases, worst abs error %.2e at %s' % (r['n'], r['worst'], r['worst_case']))
