"""Dev: the training fuzzer on ONE configuration (8 rows: B = 1, F = 8, no RNN, N = 2, hidden 64 -- soak case 6605/561) over many
seeds, with the one-launch layers on and off: distribution of (engine-vs-autograd error / tolerance)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from tests.fuzz import fuzz_train

for cols in (0, 1):
    fuzz_train.FORCE.clear()
    fuzz_train.FORCE.update(rnn=False, n_markers=12, N=2, B=1, F=8, hidden=64, fused=0, epi=0, cols=cols)
    worst, fails, flips = [], 0, 0
    for seed in range(40):
        try:
            r = fuzz_train.run(seed=7000 + seed, n_cases=1, log=lambda m: None) if 'log' in fuzz_train.run.__code__.co_varnames \
                else fuzz_train.run(7000 + seed, n_cases=1)
            worst.append(r['worst']); flips += r['flips']
        except AssertionError as e:
            fails += 1
    w = np.array(worst)
    print('train_cols=%d: %d seeds, %d refused, %d flips; err/tol median %.2f, 90%% %.2f, max %.2f'
          % (cols, 40, fails, flips, np.median(w), np.quantile(w, 0.9), w.max()))
from em_pose_amd import _lib
_lib.check(_lib.lib().empose_reset_options())
