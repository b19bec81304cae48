"""
The evaluation ENTRY POINT against the reference's own (BASELINE.json north_star: "keeping the ... evaluate_real.py entry
points so it drops in").

tests/golden/eval_assets/ is an asset tree in the reference's layout -- `<EM_EXPERIMENTS>/<id>-<name>/{config.json,
model.pth}` written by the reference's writers (the checkpoint holds the `smpl.bm.*` buffers), `<EM_DATA_REAL>/*_clean.npz`
recordings of 70 / 256 / 300 / 520 frames (+ hold_out/ 40 / 270) with missing sensors, `*_offsets.npz` -- and
expected.{json,npz} hold what the UNMODIFIED /root/reference/scripts/evaluate_real.py::main printed on it and what
empose/eval/helpers.py::evaluate returned (tests/golden/make_golden.py --only-eval-assets, build container only).

Here `python scripts/evaluate_real.py --model_id <id> [--cross_subject] --json` of THIS repository runs on the same tree
(its `--model_id` branch: config.json -> model.pth -> recordings), with both drivers, and must print the same table:
every MPJPE / PA-MPJPE row to 1e-3 mm, every MPJAE row to 1e-3 degrees.
"""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(H.GOLDEN, 'eval_assets')
CASES = ['lgdrnn6_test_real', 'lgdrnn6_hold_out', 'lgdrnn12_hold_out']
TOL_ROW = 1e-3   # mm and degrees


def expected():
    with open(os.path.join(ASSETS, 'expected.json')) as f:
        return json.load(f)


def asset_env(tmp_path, **extra):
    """The four directories the reference reads from the environment (configuration.py:25-28).  The body model of the
    tree is the 160-vertex stand-in all goldens use (tests/golden/smpl_small.npz), copied to where
    `create_default_smpl_model` looks for the licensed one."""
    d = os.path.join(str(tmp_path), 'smpl_models', 'smplh_amass', 'neutral')
    os.makedirs(d, exist_ok=True)
    shutil.copy(os.path.join(H.GOLDEN, 'smpl_small.npz'), os.path.join(d, 'model.npz'))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(EM_EXPERIMENTS=os.path.join(ASSETS, 'experiments'), EM_DATA_REAL=os.path.join(ASSETS, 'data_real'),
               SMPL_MODELS=os.path.join(str(tmp_path), 'smpl_models'), EM_DATA_SYNTH=str(tmp_path))
    env.update(extra)
    return env


def run_cli(env, e, *flags):
    cmd = [sys.executable, os.path.join(ROOT, 'scripts', 'evaluate_real.py'), '--model_id', str(e['model_id']), '--json']
    cmd += (['--cross_subject'] if e['cross_subject'] else []) + list(flags)
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]), r.stdout


def check_rows(got, e):
    assert got['headers'] == e['headers'][2:]
    assert [r[:2] for r in got['rows']] == [r[:2] for r in e['rows']]           # Nr, recording id / 'Overall average'
    for r, w in zip(got['rows'], e['rows']):
        np.testing.assert_allclose(r[2:], w[2:], atol=TOL_ROW, rtol=0, err_msg=str(r[1]))


def test_asset_tree_is_what_the_cli_reads(tmp_path, monkeypatch):
    """CPU: the `--model_id` branch's host half -- directory lookup, config.json, checkpoint incl. `smpl.bm.*`, sensor
    sites from the offsets files, recordings and their normalisation -- on the reference-written tree."""
    for k, v in asset_env(tmp_path).items():
        if k.startswith(('EM_', 'SMPL_')):
            monkeypatch.setenv(k, v)
    from em_pose_amd.data.data import RealSample
    from em_pose_amd.eval import helpers as EH
    from em_pose_amd.helpers.configuration import CONSTANTS as C
    from oracle import eval_ref as E
    exp = expected()
    assert EH.sensor_vertex_ids() == exp['vertex_ids'] != list(C.VERTEX_IDS)
    net, cfg, model_dir = EH.load_model(1615200973, torch.device('cpu'))
    assert os.path.basename(model_dir) == '1615200973-IEF-2x32-N4-RNN-2x32-r0.01-ws32-lr0.001-grad-n12-pos-ori'
    assert net.vertex_ids == exp['vertex_ids'] and net.N == 4 and net.n_markers == 12 and not net.training
    sd = torch.load(os.path.join(model_dir, 'model.pth'), map_location='cpu')['model_state_dict']
    mine = net.state_dict()
    for k, v in sd.items():      # every tensor of the reference's checkpoint, body-model buffers included, has landed
        assert k in mine and torch.equal(mine[k].cpu(), v), k
    # the recordings, normalised: equal to the oracle's restatement of the reference's loader (pinned by expected.json)
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import evaluate_real as cli
    for f in sorted(os.listdir(os.path.join(ASSETS, 'data_real'))):
        if not f.endswith('_clean.npz'):
            continue
        path = os.path.join(ASSETS, 'data_real', f)
        b = cli.sample_to_batch(RealSample.from_npz_clean(path))
        want = E.load_recording(path)
        assert b.ids == [want['id']]
        np.testing.assert_allclose(b.marker_pos_real[0].numpy(), want['marker_pos'].numpy(), atol=1e-6)
        np.testing.assert_allclose(b.marker_ori_real[0].numpy(), want['marker_oris'].numpy(), atol=1e-6)
        np.testing.assert_allclose(b.poses[0].numpy(), E.normalize_root(want['poses']).numpy(), atol=1e-6)
        assert (b.marker_masks[0].numpy() == want['masks'].numpy()).all()


def test_checkpoint_body_model_buffers_replace_the_npz(tmp_path):
    """CPU: a checkpoint whose `smpl.bm.*` differ from model.npz -- the network must compute with the checkpoint's, as the
    reference's does after load_state_dict (eval/helpers.py:131-137)."""
    from em_pose_amd.bodymodels.smpl import SMPLLayer
    model = H.small_model()
    layer = SMPLLayer(model)
    v0 = layer.tables_version
    sd = layer.state_dict()
    layer.load_state_dict(sd)
    assert layer.tables_version == v0                     # identical buffers: nothing to rebuild
    sd = {k: v.clone() for k, v in sd.items()}
    sd['bm.v_template'] += 0.01
    sd['bm.posedirs'] *= 2.0
    layer.load_state_dict(sd)
    assert layer.tables_version == v0 + 1
    np.testing.assert_allclose(layer.model['v_template'], model['v_template'] + 0.01, atol=1e-7)
    np.testing.assert_allclose(layer.model['posedirs'], 2.0 * model['posedirs'], atol=1e-9)
    assert layer.model['posedirs'].shape == model['posedirs'].shape


@pytest.mark.gpu
@pytest.mark.parametrize('tag', CASES)
@pytest.mark.parametrize('driver', ['batched', 'sequential'])
def test_cli_reproduces_the_reference_table(tmp_path, tag, driver):
    e = expected()[tag]
    got, out = run_cli(asset_env(tmp_path), e, *(['--sequential'] if driver == 'sequential' else []))
    check_rows(got, e)
    assert got['frames'] == ({False: 70 + 256 + 300 + 520, True: 40 + 270}[e['cross_subject']])
    assert 'E2E {}'.format(e['model_id']) in out          # the table header of evaluate_real.py:99-100


@pytest.mark.gpu
def test_cli_two_ranks_give_the_same_table(tmp_path):
    """Recordings sharded over two self-spawned ranks (one GPU here: EMPOSE_SHARE_DEVICES wraps them around it, gloo
    carries the metric gather), rows gathered: the table of the reference again."""
    e = expected()['lgdrnn6_test_real']
    env = asset_env(tmp_path, EMPOSE_SHARE_DEVICES='1', EMPOSE_DIST_BACKEND='gloo')
    got, _ = run_cli(env, e, '--gpus', '2')
    assert got['n_gpus'] == 2
    check_rows(got, e)


@pytest.mark.gpu
def test_per_chunk_outputs_and_library_evaluate_match_the_reference(tmp_path, monkeypatch):
    """In process: (1) every 256-frame chunk's pose / root / shape estimate, state carried chunk to chunk, equals what the
    reference's model returned inside its `main()`; (2) `em_pose_amd.eval.helpers.evaluate` (reference eval/helpers.py:
    51-111: losses through `net.backward` + metrics) returns the reference's loss values and metrics."""
    for k, v in asset_env(tmp_path).items():
        if k.startswith(('EM_', 'SMPL_')):
            monkeypatch.setenv(k, v)
    from em_pose_amd.bodymodels.smpl import create_default_smpl_model
    from em_pose_amd.data.data import RealSample
    from em_pose_amd.data.transforms import get_end_to_end_preprocess_fn
    from em_pose_amd.eval import helpers as EH
    from em_pose_amd.eval.metrics import MetricsEngine
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import evaluate_real as cli
    exp, z = expected(), np.load(os.path.join(ASSETS, 'expected.npz'))
    dev = torch.device('cuda:0')
    for tag in CASES:
        e = exp[tag]
        net, cfg, _ = EH.load_model(e['model_id'], dev)
        base = os.path.join(ASSETS, 'data_real', 'hold_out' if e['cross_subject'] else '')
        files = sorted(f for f in os.listdir(base) if f.endswith('_clean.npz'))
        for s, f in enumerate(files):
            batch = cli.sample_to_batch(RealSample.from_npz_clean(os.path.join(base, f)))
            for c, chunk in enumerate(EH.window_generator(batch, 256)):
                out = net(chunk.to_gpu(dev), is_new_sequence=(c == 0))
                for k in ('pose_hat', 'root_ori_hat', 'shape_hat'):
                    np.testing.assert_allclose(out[k].cpu().numpy(), z['{}/seq{}/chunk{}/{}'.format(tag, s, c, k)],
                                               atol=1e-4, rtol=0, err_msg='{} {} chunk {} {}'.format(tag, f, c, k))
    # (2) the library loop
    e = exp['evaluate_lgdrnn6_test_real']
    net, cfg, _ = EH.load_model(1615631737, dev)
    smpl = create_default_smpl_model(dev)
    pre = get_end_to_end_preprocess_fn(cfg, smpl, list(EH.get_all_offset_files().values()))
    base = os.path.join(ASSETS, 'data_real')
    from em_pose_amd.data.data import RealBatch
    from em_pose_amd.data.transforms import NormalizeRealMarkers, ToTensor
    loader = [RealBatch.from_sample_list([ToTensor()(NormalizeRealMarkers()(RealSample.from_npz_clean(
        os.path.join(base, f))))]) for f in sorted(os.listdir(base)) if f.endswith('_clean.npz')]
    me = MetricsEngine(smpl)
    losses = EH.evaluate(loader, net, pre, me, window_size=256, device=dev)
    for k, v in e['losses'].items():
        assert losses[k] == pytest.approx(v, rel=2e-5, abs=1e-6), k
    for k, v in e['metrics'].items():
        assert me.get_metrics()[k] == pytest.approx(v, abs=TOL_ROW), k
