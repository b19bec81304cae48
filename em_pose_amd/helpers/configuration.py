"""
Constants and run configuration for the LGD path.

Mirrors the data (not the code) of the reference's `empose/helpers/configuration.py`:
  * VERTEX_IDS            reference configuration.py:32-34
  * S_CONFIG_6            reference configuration.py:89
  * N_JOINTS / N_JOINTS_HAND / N_SHAPE_PARAMS   reference configuration.py:104-107
  * SMPL_PARENTS          reference configuration.py:118
  * flag names/defaults   reference configuration.py:147-212

Differences on purpose: the four data-path environment variables are optional here (the reference raises KeyError
at import when they are unset, configuration.py:25-28) because the hot path does not need them; they are only
consulted by `create_default_smpl_model` and the `evaluate_real` CLI.
"""
import argparse
import json
import os
import pprint

import torch


class _Constants(object):
    def __init__(self):
        self.DTYPE = torch.float32
        self.FPS = 60.0

        # Virtual sensor sites on the SMPL mesh (vertex ids), in network sensor order.
        self.VERTEX_IDS = [3027, 3748,
                           5430, 5178, 5006, 4447, 4559,
                           1961, 1391, 1535, 959, 1072]
        # Sensor names in the order the networks expect them (back, head, right arm/leg chain, left arm/leg chain).
        self.S_ORDER = ['ID120.Set7.Num8', 'ID113.Set7.Num1',
                        'ID115.Set7.Num3', 'ID117.Set7.Num5', 'ID119.Set7.Num7', 'ID121.Set7.Num9',
                        'ID123.Set7.Num11',
                        'ID114.Set7.Num2', 'ID116.Set7.Num4', 'ID118.Set7.Num6', 'ID122.Set7.Num10',
                        'ID124.Set7.Num12']
        # The reduced 6-sensor configuration: back, head, both wrists, both lower legs.
        self.S_CONFIG_6 = [0, 1, 2, 6, 7, 11]
        self.N_TRACKERS_WO_ROOT = 12

        # SMPL-H constants.
        self.N_JOINTS = 21  # body joints, root not counted
        self.MAX_INDEX_ROOT_AND_BODY = 66
        self.N_JOINTS_HAND = 15
        self.N_SHAPE_PARAMS = 10
        self.SMPL_JOINTS = ['root', 'l_hip', 'r_hip', 'spine1', 'l_knee', 'r_knee', 'spine2', 'l_ankle', 'r_ankle',
                            'spine3', 'l_foot', 'r_foot', 'neck', 'l_collar', 'r_collar', 'head', 'l_shoulder',
                            'r_shoulder', 'l_elbow', 'r_elbow', 'l_wrist', 'r_wrist']
        self.SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19]

    @property
    def DEVICE(self):
        return torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')

    # Paths are looked up lazily so that importing the package never fails.
    @property
    def DATA_DIR(self):
        return os.environ.get('EM_DATA_SYNTH', '')

    @property
    def EXPERIMENT_DIR(self):
        return os.environ.get('EM_EXPERIMENTS', '')

    @property
    def SMPL_MODELS_DIR(self):
        return os.environ.get('SMPL_MODELS', '')

    @property
    def DATA_DIR_TEST(self):
        return os.environ.get('EM_DATA_REAL', '')


CONSTANTS = _Constants()


# (flag, kwargs) pairs; same names and defaults as the reference CLI so that `config.json` files written by the
# reference load unchanged and `cmd.txt` lines can be replayed.
_FLAGS = [
    ('--experiment_id', dict(default=None)),
    ('--seed', dict(type=int, default=None)),
    ('--data_workers', dict(type=int, default=4)),
    ('--print_every', dict(type=int, default=25)),
    ('--eval_every', dict(type=int, default=700)),
    ('--tag', dict(default='')),
    ('--test', dict(action='store_true')),
    ('--m_type', dict(default='rnn', choices=['rnn', 'resnet', 'ief', 'lgd'])),
    ('--m_estimate_shape', dict(action='store_true')),
    ('--m_shape_hidden_size', dict(default=256)),
    ('--m_fk_loss', dict(type=float, default=0.0)),
    ('--m_dropout', dict(type=float, default=0.0)),
    ('--m_hidden_size', dict(type=int, default=1024)),
    ('--m_num_layers', dict(type=int, default=2)),
    ('--m_learn_init_state', dict(action='store_true')),
    ('--m_bidirectional', dict(action='store_true')),
    ('--m_num_iterations', dict(type=int, default=4)),
    ('--m_dropout_hidden', dict(type=float, default=0.0)),
    ('--m_step_size', dict(type=float, default=0.1)),
    ('--m_reprojection_loss_weight', dict(type=float, default=0.01)),
    ('--m_shape_loss_weight', dict(type=float, default=1.0)),
    ('--m_pose_loss_weight', dict(type=float, default=1.0)),
    ('--m_average_shape', dict(action='store_true')),
    ('--m_use_gradient', dict(action='store_true')),
    ('--m_skip_connections', dict(action='store_true')),
    ('--m_no_batch_norm', dict(action='store_true')),
    ('--m_rnn_init', dict(action='store_true')),
    ('--m_rnn_denoiser', dict(action='store_true')),
    ('--m_rnn_bidirectional', dict(action='store_true')),
    ('--m_rnn_hidden_size', dict(type=int, default=512)),
    ('--m_rnn_num_layers', dict(type=int, default=2)),
    ('--use_marker_pos', dict(action='store_true')),
    ('--use_marker_ori', dict(action='store_true')),
    ('--use_marker_nor', dict(action='store_true')),
    ('--use_real_offsets', dict(action='store_true')),
    ('--offset_noise_level', dict(type=int, default=0)),
    ('--n_markers', dict(type=int, default=12)),
    ('--noise_num_markers', dict(type=int, default=1)),
    ('--spherical_noise_strength', dict(type=float, default=0.0)),
    ('--spherical_noise_length', dict(type=float, default=0.0)),
    ('--suppression_noise_length', dict(type=float, default=0.0)),
    ('--suppression_noise_value', dict(type=float, default=0.0)),
    ('--lr', dict(type=float, default=0.001)),
    ('--n_epochs', dict(type=int, default=50)),
    ('--bs_train', dict(type=int, default=16)),
    ('--bs_eval', dict(type=int, default=16)),
    ('--eval_window_size', dict(type=int, default=None)),
    ('--window_size', dict(type=int, default=120)),
    ('--load', dict(action='store_true')),
]


class Configuration(object):
    """Flat attribute bag; JSON round-trips with the reference's `config.json` (configuration.py:214-225)."""

    def __init__(self, adict):
        self.__dict__.update(adict)

    def __str__(self):
        return pprint.pformat(vars(self), indent=4)

    @staticmethod
    def _parser():
        p = argparse.ArgumentParser()
        for flag, kw in _FLAGS:
            p.add_argument(flag, **kw)
        return p

    @staticmethod
    def defaults(**overrides):
        """All flags at their defaults, then `overrides` applied (programmatic construction)."""
        cfg = vars(Configuration._parser().parse_args([]))
        unknown = set(overrides) - set(cfg)
        if unknown:
            raise ValueError('Unknown configuration keys: {}'.format(sorted(unknown)))
        cfg.update(overrides)
        return Configuration(cfg)

    @staticmethod
    def parse_cmd(argv=None):
        return Configuration(vars(Configuration._parser().parse_args(argv)))

    @staticmethod
    def from_json(json_path):
        with open(json_path, 'r') as f:
            return Configuration(json.load(f))

    def to_json(self, json_path):
        with open(json_path, 'w') as f:
            f.write(json.dumps(vars(self), indent=2, sort_keys=True))


def lgd_config(n_markers=12, rnn=True, num_iterations=4, window_size=32, hidden=512, rnn_hidden=512, lr=0.001,
               **more):
    """The flag set of the released LGD / LGD-RNN models (reference README.md:221, model names README.md:51,229)."""
    kw = dict(m_type='ief', m_hidden_size=hidden, m_num_layers=2, m_num_iterations=num_iterations,
              window_size=window_size, use_marker_pos=True, use_marker_ori=True, use_real_offsets=True,
              offset_noise_level=0, m_average_shape=True, m_use_gradient=True, m_reprojection_loss_weight=0.01,
              eval_window_size=256, m_rnn_init=bool(rnn), m_rnn_hidden_size=rnn_hidden, lr=lr,
              n_markers=n_markers, m_pose_loss_weight=10.0, m_fk_loss=0.1)
    kw.update(more)
    return Configuration.defaults(**kw)
