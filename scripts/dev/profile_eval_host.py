"""Dev: cProfile of the one-recording-at-a-time evaluation driver (host time per Python function, second pass)."""
import cProfile, pstats, sys, io
sys.path.insert(0, '.'); sys.path.insert(0, 'scripts')
import torch
import evaluate_real as E
from types import SimpleNamespace
args = SimpleNamespace(model_id=1615631737, n_markers=6, iterations=2, no_rnn=False, m_type='ief', max_sequences=0)
dev = torch.device('cuda:0')
net, smpl, lengths, load, name = E.synthetic_setup(args, dev)
batches = [load(i) for i in range(len(lengths))]
net.keep_history = False
from em_pose_amd.eval.helpers import evaluate_sequences
evaluate_sequences(net, batches, smpl, dev, window_size=256)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
evaluate_sequences(net, batches, smpl, dev, window_size=256)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
