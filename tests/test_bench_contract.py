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


def test_bench_multi_gpu_needs_the_launcher():
    r = _run(['--gpus', '2'], env={'WORLD_SIZE': '1'})
    assert r.returncode != 0
    assert 'torch.distributed.run' in (r.stderr + r.stdout)


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
def test_graft_entry_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()
