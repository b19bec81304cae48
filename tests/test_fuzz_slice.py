"""A fixed-seed slice of the randomized checks under tests/fuzz/, inside `pytest -m gpu` (round-3 lesson: the only
memory-safety bug of that round, an LDS overrun of `heads_rows_kernel` with narrow LSTMs, was found by a fuzzer that
pytest never ran).  Same generators as the command-line fuzzers, fixed seeds, fixed case counts; every case above 1e-5
is logged with its kernel variants.  Reference semantics being protected: reference empose/nn/models.py:485-632,
empose/nn/layers.py:133-157 (RNNLayer), empose/nn/layers.py:13-77 (MLP)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _explained_by_conditioning(f64):
    """A case above 1e-5 is accepted as conditioning -- not a kernel's inaccuracy -- when the HIP outputs are no further
    from the float64 oracle than (a) twice what the fp32 oracle is, or (b) four times what the float64 outputs themselves
    move (the larger of two draws) when the body-model constants and the inputs move by ONE fp32 unit in the last place (the noise every fp32
    evaluation of the vertices commits; the rule of the training goldens, tests/golden/train_sensitivity.json).
    Round 4 replayed the B=257 / F=3 case stage by stage (scripts/dev/replay_case20.py, profiles/r04_replay_case20_*.txt):
    one frame -- the only valid frame of its window, so its gradient is scaled by F / n_frames = 3 -- has a sensor whose
    estimated orientation is within rounding of its target (the residual direction r / |r| is then decided by the last
    bits of the vertices): orientation error 1e-5 already in the FORWARD pass of both implementations, gradient 675
    against a typical 40.  Over the other 287 frames the per-frame gradient errors of HIP and of the fp32 oracle have the
    same distribution (median ratio 0.97-1.04); one-ulp noise on the model constants moves the float64 outputs by 2.6e-5,
    one-ulp noise on the inputs alone by 2.4e-7: the error belongs to fp32 vertices, not to a kernel."""
    return f64['hip_vs_f64'] <= max(1e-5, 2.0 * f64['oracle_f32_vs_f64'], 4.0 * f64['one_ulp_sensitivity'])


def _show(capsys, text):
    with capsys.disabled():
        print('\n[fuzz slice] ' + text, flush=True)


def test_lgd_forward_slice_all_kernel_variants_narrow_and_wide_nets(capsys):
    """120 LGD / LGD-RNN forwards: golden nets + random-init nets with LSTMs of 8 / 16 / 64 units next to update nets of
    16..128 units (hidden < input and LSTM < heads' staging tile included), B in {1..257}, ragged lengths, missing sensors
    (host- or device-side suppression), carried state, all 16 combinations of (frame-per-lane SMPL kernels, fused blend
    GEMMs, on-device suppression, two-part forward on two streams).  The oracle's outputs of these fixed-seed cases were
    recorded in the build container (tests/golden/fuzz_lgd_slice_4101.npz: 512 seeded entries + 4 projections per output
    and case; `python tests/fuzz/fuzz_lgd.py record`), so the GPU box no longer spends 0.9 s of host time per case on the
    oracle (round 4 stopped at 48 cases for that reason); the first 12 cases ALSO run the oracle live and compare every
    entry, which ties the fixture to the oracle on this box."""
    import os
    from tests.fuzz import fuzz_lgd
    cache = fuzz_lgd.load_fixture(fuzz_lgd.slice_fixture_path()) if os.path.exists(fuzz_lgd.slice_fixture_path()) else None
    assert cache is not None and len(cache) == fuzz_lgd.SLICE_CASES
    live = fuzz_lgd.run(seed=fuzz_lgd.SLICE_SEED, n_cases=12, extra_nets=True, batches=fuzz_lgd.BATCHES_SLICE,
                        log=lambda m: _show(capsys, m))
    r = fuzz_lgd.run(seed=fuzz_lgd.SLICE_SEED, n_cases=fuzz_lgd.SLICE_CASES, extra_nets=True,
                     batches=fuzz_lgd.BATCHES_SLICE, oracle_cache=cache, log=lambda m: _show(capsys, m))
    _show(capsys, 'lgd: %d cases against recorded oracle fingerprints, worst %.2e at %s; %d above 1e-5; first 12 live '
          'against the oracle, every entry: worst %.2e; variants %s'
          % (r['n'], r['worst'], r['worst_case'], len(r['above_1e5']), live['worst'], sorted(r['variants'].items())))
    assert r['n'] == fuzz_lgd.SLICE_CASES and r['worst'] < 1e-4 and live['n'] == 12 and live['worst'] < 1e-4
    # the sampled comparison of a case sees (almost) what the full one sees
    for a, b in zip(live['errors'], r['errors'][:12]):
        assert b <= a + 1e-7 and b >= 0.2 * a - 1e-7, (a, b)
    assert len(r['variants']) >= 12     # the slice does visit the variant combinations
    # errors of a few 1e-5 are input conditioning when they occur (see the regression below)
    for case, err, desc, f64 in r['above_1e5'] + live['above_1e5']:
        assert _explained_by_conditioning(f64), (case, err, desc, f64)


@pytest.mark.parametrize('force', [(0, 0, 1, 1), (2, 1, 1, 1)], ids=['general_kernels', 'frame_per_lane'])
def test_regression_short_masked_carried_windows_b257_f3(force, capsys):
    """The one case above 1e-5 in 700 (seed 3602, case 20: 257 windows of 3 frames, missing sensors, carried LSTM state;
    round 3 measured 1.57e-5 on the general kernels and 1.34e-5 on the frame-per-lane ones).  Pinned with its measured
    bound, and with its conditioning: against the float64 oracle the HIP path is no worse than the fp32 oracle is."""
    from em_pose_amd import _lib
    from tests.fuzz import fuzz_lgd
    # round 5: at 257 rows the LSTM steps now form their products from bf16 pieces (lstm_x3.hip), whose rounding differs --
    # the default path lands BELOW 1e-5 on this case; the pinned regression is the fp32-MFMA kernels it was found on
    r = fuzz_lgd.run(seed=3602, n_cases=21, start=20, force=list(force), log=lambda m: _show(capsys, m))
    assert r['n'] == 1 and r['worst'] < 2.5e-5, r['worst']
    for case, err, desc, f64 in r['above_1e5']:
        assert _explained_by_conditioning(f64), f64
    # (all three: the update nets, the LSTM steps and the row-block SMPL products)
    pinned = (b'lstm_x3', b'mlp_x3', b'rows_x3')
    for name in pinned:
        _lib.check(_lib.lib().empose_set_option(name, 0))
    try:
        r = fuzz_lgd.run(seed=3602, n_cases=21, start=20, force=list(force), log=lambda m: _show(capsys, m))
    finally:
        for name in pinned:
            _lib.check(_lib.lib().empose_set_option(name, 1))
    assert r['n'] == 1 and r['worst_case'][1] == 'lgdrnn12_n4_carry'
    assert r['worst_case'][2] == dict(B=257, F=3, masks=True, state=True)
    assert r['worst'] < 2.5e-5, r['worst']
    assert r['above_1e5'], 'the case is expected above 1e-5 (if a change made it better, tighten the bound above)'
    for case, err, desc, f64 in r['above_1e5']:
        assert _explained_by_conditioning(f64), f64


def test_lstm_slice(capsys):
    """150 random LSTM stacks vs torch.nn.LSTM on packed sequences: 1-4 layers, uni / bidirectional, 4..64 units,
    B in {1..700} (all batch regimes + the opt-in whole-sequence kernel), ragged lengths, given state."""
    from tests.fuzz import fuzz_lstm
    r = fuzz_lstm.run(seed=4102, n_cases=150, log=lambda m: _show(capsys, m))
    _show(capsys, 'lstm: %d cases, worst %.2e at %s' % (r['n'], r['worst'], r['worst_case']))
    assert r['n'] == 150 and r['worst'] < 1e-4


def test_linear_and_mesh_slice(capsys):
    """300 random linear layers (every GEMM tile regime, bias / PReLU / residual) vs float64; 40 full-mesh evaluations
    x (fp32, split-bf16) x both Rodrigues conventions vs the oracle."""
    from tests.fuzz import fuzz_linear_mesh as F
    r = F.run_linear(seed=4103, n_cases=300)
    _show(capsys, 'linear: %d cases, worst %.2e at %s' % (r['n'], r['worst'], r['worst_case']))
    assert r['n'] == 300
    m = F.run_mesh(seed=4104, n_cases=40)
    _show(capsys, 'mesh: %d cases, worst %.2e (f32) / %.2e (bf16x3)' % (m['n'], m['worst']['f32'], m['worst']['bf16x3']))
    assert m['n'] == 40 and max(m['worst'].values()) < 3e-5


def test_training_step_slice(capsys):
    """60 random training configurations: the engine's hand-written reverse sweep vs autograd over the same kernels,
    every parameter gradient within 2e-4 of scale + 8x the step's own one-ulp sensitivity.  (A property test -- two
    derivations of one backward; parity with the reference is tests/test_hip_parity.py::test_training_step_matches_
    reference_gradients and tests/test_hip_round5.py::test_full_width_training_step_matches_reference_fingerprints.)"""
    from tests.fuzz import fuzz_train
    r = fuzz_train.run(seed=4105, n_cases=60)
    _show(capsys, 'train: %d configurations, worst gradient error / tolerance %.2f, %d PReLU kink flips'
          % (r['n'], r['worst'], r['flips']))
    # `worst` is over every compared gradient tensor; a configuration with a PReLU kink flip (accepted only by its
    # footprint, tests/fuzz/fuzz_train.py::_kink_flip_pattern) contributes its unaffected tensors
    assert r['n'] == 60 and r['flips'] <= 6 and r['worst'] <= 1.0
