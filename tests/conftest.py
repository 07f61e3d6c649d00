import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


DP_RESULT = os.path.join(ROOT, 'gpurun_out', 'dp_gpu_result.json')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """The 2-rank data-parallel run of the HIP path (tests/dp_gpu_worker.py, checked by tests/test_dp_gpu.py) is a set of
    child processes.  It is started HERE, before anything in this pytest process has initialised HIP: on the GPU pool a
    process that has touched the GPU must not exec another program, so the children are launched while this process
    is still GPU-free (torch.cuda.device_count() does not initialise the device on this image) and the test later only
    reads the result file."""
    expr = session.config.getoption('markexpr') or ''
    if 'gpu' not in expr or 'not gpu' in expr:
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    import subprocess
    os.makedirs(os.path.dirname(DP_RESULT), exist_ok=True)
    if os.path.exists(DP_RESULT):
        os.remove(DP_RESULT)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    log = open(DP_RESULT + '.log', 'w')
    try:
        subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dp_gpu_worker.py'), DP_RESULT], env=env, stdout=log,
                       stderr=subprocess.STDOUT, timeout=900)
    except Exception as e:  # the test reports the missing result
        log.write('launcher: %r\n' % (e,))
    finally:
        log.close()


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('GPU test selected but no GPU is visible')
    return torch.device('cuda:0')
