"""bench.py / __graft_entry__ contract: one JSON line with the fields the driver reads; loud failure without a GPU."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=e,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU behaviour')
def test_bench_refuses_to_run_without_a_gpu():
    r = _run(['--steps', '1', '--warmup', '0'])
    assert r.returncode != 0
    assert 'no CPU fallback' in (r.stderr + r.stdout)


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 2, reason='checks the too-few-GPUs exit')
def test_bench_multi_gpu_says_how_many_gpus_it_found():
    """`python bench.py --gpus N` is a plain command (no torch.distributed.run needed): with fewer than N devices it
    exits non-zero and says so."""
    r = _run(['--gpus', '2'], env={'WORLD_SIZE': '1'})
    assert r.returncode != 0
    assert 'needs 2 GPUs, found' in (r.stderr + r.stdout)


@pytest.mark.skipif(torch.cuda.is_available(), reason='drives the launcher up to the point where a device is needed')
@pytest.mark.parametrize('script,needle', [('bench.py', 'no CPU fallback'),
                                           (os.path.join('scripts', 'evaluate_real.py'), 'no CPU fallback'),
                                           (os.path.join('scripts', 'train.py'), 'no CPU fallback')])
def test_plain_command_spawns_its_own_ranks(script, needle):
    """--gpus 2 without a launcher: two rank processes are spawned, rendezvous over 127.0.0.1 (gloo here: the
    EMPOSE_DIST_BACKEND switch skips the device count), pass a barrier and only then stop at the device check."""
    env = dict(os.environ, EMPOSE_DIST_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    extra = ['--synthetic'] if 'evaluate_real' in script else []
    r = subprocess.run([sys.executable, os.path.join(ROOT, script), '--gpus', '2'] + extra, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    text = r.stderr + r.stdout
    assert r.returncode != 0
    assert 'rank 0/2 joined the process group over gloo' in text, text[-2000:]
    assert 'rank 1/2 joined the process group over gloo' in text, text[-2000:]
    assert needle in text


def test_launch_ranks_returns_the_failing_status_and_stops_the_others(tmp_path):
    from em_pose_amd.helpers.distributed import launch_ranks
    script = tmp_path / 'ranks.py'
    script.write_text('import os, sys, time\n'
                      'r = int(os.environ["RANK"]); assert os.environ["WORLD_SIZE"] == "3"\n'
                      'assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0\n'
                      'open(sys.argv[1] + "/r%d" % r, "w").write(os.environ["LOCAL_RANK"])\n'
                      'if sys.argv[2] == "fail" and r == 1: sys.exit(7)\n'
                      'if sys.argv[2] == "fail": time.sleep(120)\n')
    assert launch_ranks(str(script), [str(tmp_path), 'ok'], 3) == 0
    assert sorted(p.name for p in tmp_path.glob('r*') if p.name != 'ranks.py') == ['r0', 'r1', 'r2']
    import time
    t0 = time.time()
    assert launch_ranks(str(script), [str(tmp_path), 'fail'], 3) == 7
    assert time.time() - t0 < 60


@pytest.mark.gpu
def test_bench_force_dist_goes_through_the_spawn_path_and_agrees_with_the_plain_run():
    """`bench.py --gpus 1 --force_dist`: the SAME self-launch + RCCL process-group path `--gpus 8` takes, on one GPU;
    its value agrees with the plain single-process run."""
    common = ['--steps', '10', '--warmup', '3', '--no_cpu_baseline', '--no_traffic']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    runs = {}
    for name, extra in (('plain', []), ('spawn', ['--gpus', '1', '--force_dist'])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + common + extra, cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        assert len(lines) == 1
        runs[name] = json.loads(lines[0])
    assert runs['spawn']['n_gpus'] == 1 and runs['spawn']['config']['process_group'] == 'nccl'
    assert runs['plain']['config']['process_group'] is None
    assert runs['spawn']['value'] == pytest.approx(runs['plain']['value'], rel=0.02)


@pytest.mark.gpu
@pytest.mark.parametrize('script,extra,key', [
    (os.path.join('scripts', 'evaluate_real.py'), ['--synthetic', '--max_sequences', '3', '--json'], 'frames_per_sec'),
    (os.path.join('scripts', 'train.py'), ['--steps', '3', '--warmup', '1', '--bs_train', '4', '--json'], 'frames_per_sec')])
def test_eval_and_train_scripts_spawn_their_own_rank_with_rccl(script, extra, key):
    """configs[3] / [4] entry points: `--gpus 1 --force_dist` is the path `--gpus 8` takes (self-launch, RCCL group, the
    metric gather / the gradient buckets), on the one GPU of this box."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, script), '--gpus', '1', '--force_dist'] + extra, cwd=ROOT,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d[key] > 0
    if 'train' in script:
        assert d['process_group'] == 'nccl' and d['gradient_collectives_per_step'] >= 1


@pytest.mark.gpu
def test_training_with_buckets_rccl_and_side_streams_equals_the_plain_step():
    """64 windows = 2048 frames per step: the engine's side streams are on; with `--gpus 1 --force_dist` the gradients go
    through the persistent buckets and RCCL (a bucket whose members become final on two different side streams waits for
    both).  One rank: averaging changes nothing, so every loss value of every step equals the plain run's."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    outs = []
    for extra in ([], ['--gpus', '1', '--force_dist']):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'train.py'), '--steps', '3', '--warmup', '1',
                            '--bs_train', '64'] + extra, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append([ln.split(' elapsed')[0] for ln in r.stdout.splitlines() if ln.startswith('[TRAIN')])
    assert len(outs[0]) == 4 and outs[0] == outs[1]


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_run_the_multi_rank_logic_on_device_results():
    """No second GPU on this box: EMPOSE_SHARE_DEVICES=1 wraps the ranks around the devices there are and
    EMPOSE_DIST_BACKEND=gloo carries the collectives (RCCL refuses two ranks on one device).  Everything else is the N > 1
    path as the driver launches it: self-spawned ranks, window / sequence shards, MAX over ranks of the timed region, the
    checksum gather, the metric gather, ONE result line from rank 0."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(EMPOSE_SHARE_DEVICES='1', EMPOSE_DIST_BACKEND='gloo')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--batch', '64', '--no_cpu_baseline', '--no_traffic', '--no_profile'], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['process_group'] == 'gloo' and d['scaling'] == 'weak'
    assert d['value'] == pytest.approx(2 * 64 * 32 / (d['ms_per_step'] * 1e-3), rel=1e-6)   # whole-job frames / max-over-ranks time

    common = [sys.executable, os.path.join(ROOT, 'scripts', 'evaluate_real.py'), '--synthetic', '--max_sequences', '5',
              '--json']
    one = subprocess.run(common, cwd=ROOT, env={k: v for k, v in env.items() if not k.startswith('EMPOSE_')},
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
    two = subprocess.run(common + ['--gpus', '2'], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True, timeout=900)
    assert one.returncode == 0 and two.returncode == 0, (one.stderr[-1500:], two.stderr[-1500:])
    m1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith('{')][-1])
    m2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith('{')][-1])
    assert m2['n_gpus'] == 2 and m2['frames'] == m1['frames']
    for k, v in m1['metrics'].items():      # recordings sharded over two ranks, rows gathered: the same table
        assert m2['metrics'][k] == pytest.approx(v, rel=1e-5), k

    # data-parallel training: persistent gradient buckets all-reduced (device tensors through gloo here) every step
    t = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'train.py'), '--gpus', '2', '--steps', '3', '--warmup',
                        '1', '--bs_train', '4', '--json'], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
    assert t.returncode == 0, t.stderr[-3000:]
    d = json.loads([ln for ln in t.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['process_group'] == 'gloo' and d['gradient_collectives_per_step'] >= 1
    assert d['frames_per_sec'] > 0 and 'nan' not in t.stdout.lower()


@pytest.mark.gpu
def test_bench_with_more_gpus_than_the_box_has_exits_with_the_count():
    n = torch.cuda.device_count() + 1
    r = _run(['--gpus', str(n)])
    assert r.returncode != 0 and 'needs %d GPUs, found %d' % (n, n - 1) in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    r = _run(['--steps', '2', '--warmup', '1', '--batch', '64', '--no_cpu_baseline'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1
    assert d['unit'] == 'frames/sec' and d['higher_is_better'] is True and d['scaling'] == 'weak'
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and d['vs_baseline'] is None
    assert 'workload' in d['config'] and 'model' not in d['config']
    frames = d['config']['windows_per_gpu'] * d['config']['frames_per_window']
    assert d['value'] == pytest.approx(frames / (d['ms_per_step'] * 1e-3), rel=1e-6)
    rf = d['roofline']
    assert rf['bound'] in ('hbm', 'mfma') and rf['unit'] in ('GB/s', 'TFLOP/s')
    assert rf['frac'] == pytest.approx(rf['achieved'] / rf['peak'], rel=1e-9)
    assert 0.0 < rf['frac'] < 1.0


@pytest.mark.gpu
def test_bench_secondary_object_carries_the_other_configs():
    """The headline line of `bench.py` (BASELINE configs[2] shape) carries a `secondary` object: configs[1], the full-mesh
    workload, the configs[3] stand-in under both drivers and the configs[4] step at 12 and 256 windows -- each with what it
    ran, its rate and a roofline fraction where one applies; `value` / `metric` / `config` are the headline's own."""
    r = _run(['--steps', '5', '--warmup', '2', '--no_cpu_baseline', '--no_traffic', '--no_fp32_line'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['metric'] == 'frames/sec LGD-RNN N=4 12-sensor ws=32; MPJPE vs ref (mm)'
    assert d['value'] == pytest.approx(1024 * 32 / (d['ms_per_step'] * 1e-3), rel=1e-6)
    sec = d['secondary']
    keys = ('configs1_lgd12_b256', 'vertices_t16384', 'configs3_evaluate_real_synthetic_batched',
            'configs3_evaluate_real_synthetic_sequential', 'configs4_train_step_12_windows', 'configs4_train_step_256_windows')
    for k in keys:
        assert k in sec and 'error' not in sec[k], (k, sec.get(k))
        assert 'workload' in sec[k] and sec[k]['frames_per_sec'] > 0
    for k in ('configs1_lgd12_b256', 'vertices_t16384', 'configs4_train_step_12_windows', 'configs4_train_step_256_windows'):
        rf = sec[k]['roofline']
        assert rf['bound'] in ('hbm', 'mfma') and 0.0 < rf['frac'] < 1.0
        assert rf['frac'] == pytest.approx(rf['achieved'] / rf['peak'], rel=1e-9)
    assert sec['vertices_t16384']['roofline']['bound'] == 'hbm'
    assert sec['configs3_evaluate_real_synthetic_batched']['frames'] == 54030
    assert sec['seconds_spent'] < 240


@pytest.mark.gpu
def test_graft_entry_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()


@pytest.mark.gpu
def test_train_script_on_amass_npz_writes_a_loadable_checkpoint(tmp_path):
    """scripts/train.py --amass_dir: AMASS npz sequences -> random windows -> preprocessing -> training steps ->
    validation -> `<id>-<name>/model.pth` + `config.json` in the reference's experiment layout, loadable again."""
    import numpy as np
    rng = np.random.default_rng(0)
    data = tmp_path / 'amass' / 'subject'
    data.mkdir(parents=True)
    for i in range(6):
        n = 30 + 3 * i
        np.savez(str(data / ('seq%d.npz' % i)), poses=rng.normal(0, 0.2, size=(n, 156)), betas=rng.normal(size=16),
                 trans=rng.normal(size=(n, 3)), mocap_framerate=np.array(60.0))
    sys.path.insert(0, ROOT)
    from em_pose_amd.helpers.configuration import CONSTANTS as C
    off = str(tmp_path / '0000_offsets.npz')
    np.savez(off, means=rng.normal(0, 0.02, size=(12, 3)), covs=np.tile(np.eye(3) * 1e-4, (12, 1, 1)),
             r=np.tile(np.eye(3), (12, 1, 1)), vertex_ids=np.asarray(C.VERTEX_IDS))
    exp = tmp_path / 'experiments'
    exp.mkdir()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'train.py'), '--amass_dir', str(tmp_path / 'amass'),
                        '--offset_files', off, '--experiment_dir', str(exp), '--experiment_id', '42', '--steps', '3',
                        '--eval_every', '2', '--bs_train', '2', '--window_size', '8', '--iterations', '1', '--n_epochs', '3', '--json'],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['steps'] == 3 and np.isfinite(out['best_valid_loss'])
    assert '[VALID' in r.stdout and '***' in r.stdout
    model_dir = out['model_dir']
    assert os.path.basename(model_dir).startswith('42-') and os.path.exists(os.path.join(model_dir, 'config.json'))
    from em_pose_amd import synthetic
    from em_pose_amd.bodymodels.smpl import SMPLLayer
    from em_pose_amd.eval.helpers import get_model_dir, load_model_weights
    from em_pose_amd.helpers.configuration import Configuration
    from em_pose_amd.nn.models import create_model
    assert get_model_dir(str(exp), 42) == model_dir
    cfg = Configuration.from_json(os.path.join(model_dir, 'config.json'))
    net = create_model(cfg, SMPLLayer(synthetic.make_model()))
    load_model_weights(os.path.join(model_dir, 'model.pth'), net)
    assert cfg.window_size == 8 and net.N == 1
