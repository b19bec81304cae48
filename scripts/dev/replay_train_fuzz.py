import sys
sys.path.insert(0, '.')
from tests.fuzz import fuzz_train
cols = int(sys.argv[1])
fuzz_train.FORCE.update(cols=cols)
r = fuzz_train.run(6605, n_cases=915, keep_going=True)
print('cols=%d refused %s' % (cols, [q[0] for q in r['refused']]))
