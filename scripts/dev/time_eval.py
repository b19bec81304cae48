"""Dev: wall time of the pieces of `evaluate_real --synthetic` (both drivers), each bracketed by device syncs."""
import argparse, collections, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
import torch
import evaluate_real as E
from em_pose_amd.eval import helpers as EH, metrics as EM
from em_pose_amd.data import data as D

acc = collections.defaultdict(float); cnt = collections.Counter()
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] += time.perf_counter() - t0; cnt[name] += 1
        return r
    return w

args = argparse.Namespace(model_id=1615631737, n_markers=6, iterations=2, no_rnn=False, max_sequences=0)
dev = torch.device('cuda:0')
net, smpl, lengths, load, name = E.synthetic_setup(args, dev)
batches = [load(i) for i in range(len(lengths))]
net.keep_history = False
net.forward = timed('net.forward', net.forward)
EM.MetricsEngine.compute = timed('MetricsEngine.compute', EM.MetricsEngine.compute)
D.RealBatch.to_gpu = timed('RealBatch.to_gpu', D.RealBatch.to_gpu)
D.RealBatch.get_inputs = timed('  RealBatch.get_inputs (inside forward)', D.RealBatch.get_inputs)
for mode in ('batched', 'sequential'):
    for rep in range(2):
        acc.clear(); cnt.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == 'batched': EH.evaluate_sequences_batched(net, batches, smpl, dev, window_size=256)
        else: EH.evaluate_sequences(net, batches, smpl, dev, window_size=256)
        torch.cuda.synchronize(); total = time.perf_counter() - t0
    print('%s: %.3f s total (with the timers\' syncs)' % (mode, total))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print('   %-45s %.3f s  %4d calls' % (k, v, cnt[k]))
