"""Dev: error of the frame-per-lane SMPL path against the float64 blueprint with the row-block products on the fp32 MFMA
instruction (rows_x3 = 0) and on three bf16 pieces (rows_x3 = 1), next to the general kernels, over several seeds."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from em_pose_amd import _lib, synthetic
from em_pose_amd.helpers.configuration import CONSTANTS as CONST, lgd_config
from tests.test_hip_parity import _smpl_case, build_net
from tests.test_hip_round3 import _sensors_call, _Option
model, vids = synthetic.make_model(), CONST.VERTEX_IDS
net = build_net(lgd_config(12, False, 1, hidden=32), model, vids)
handle = net._ensure_handle(torch.device('cuda:0'))
for seed in (11, 12, 13, 14, 15, 16):
    for T, F in ((96, 8), (4209, 3)):
        theta, beta, off_r, off_t, tgt, scale, ref = _smpl_case(model, vids, T, F, seed, 12)
        refs = (ref['pos'].reshape(T, -1), ref['ori'].reshape(T, -1), ref['joints'].reshape(T, -1), ref['g_theta'], ref['g_beta'])
        row = []
        for name, opts in (('general', {b'smpl_tile': 0}), ('tile fp32', {b'smpl_tile': 2, b'rows_x3': 0}), ('tile x3', {b'smpl_tile': 2, b'rows_x3': 1})):
            for k, v in opts.items():
                _lib.check(_lib.lib().empose_set_option(k, v))
            out = _sensors_call(handle, T, F, theta, beta, off_r, off_t, tgt, scale)
            _lib.lib().empose_reset_options()
            row.append('%s: pos %.1e ori %.1e g_theta %.2e g_beta %.2e' % ((name,) + tuple(np.abs(out[i] - refs[i]).max() for i in (0, 1, 3, 4))))
        print('seed %d T=%d  |g_theta| %.1f   ' % (seed, T, np.abs(ref['g_theta']).max()) + ' | '.join(row))
