import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'provokes_poll_timeout: forces polls of the cooperative LSTM kernels to give up')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _reset_kernel_variant_options(request):
    """Tests that select a kernel variant (empose_set_option) must not leak their choice into the next test -- nor a
    poll-timeout report of a cooperative LSTM kernel (sticky since round 5: it would make every later recurrence of the
    process fail).  A test that did not provoke one on purpose (marker `provokes_poll_timeout`) fails if it leaves one."""
    yield
    from em_pose_amd import _lib
    if _lib._lib is not None:
        _lib._lib.empose_reset_options()
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            status = _lib._lib.empose_async_status()      # reports and clears
            if status != 0 and request.node.get_closest_marker('provokes_poll_timeout') is None:
                pytest.fail('a cooperative LSTM kernel of this test gave up on a poll (outputs NaN): '
                            + _lib._lib.empose_last_error().decode())
