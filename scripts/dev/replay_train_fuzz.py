"""Dev: replay the training fuzzer's cases 0 .. n-1 of a seed with an option forced in every case (tests/fuzz/fuzz_train.py::FORCE
overrides after the draws, so the sequence of cases stays what it is):  python scripts/dev/replay_train_fuzz.py <seed> <n> cols=0"""
import sys
sys.path.insert(0, '.')
from tests.fuzz import fuzz_train
seed, n = int(sys.argv[1]), int(sys.argv[2])
for kv in sys.argv[3:]:
    k, v = kv.split('=')
    fuzz_train.FORCE[k] = int(v)
r = fuzz_train.run(seed, n_cases=n, keep_going=True)
print('forced %s: refused %s, flips %d, worst %.2f' % (dict(fuzz_train.FORCE), [q[0] for q in r['refused']], r['flips'], r['worst']))
