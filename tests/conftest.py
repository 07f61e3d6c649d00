import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


DP_RESULT = os.path.join(ROOT, 'gpurun_out', 'dp_gpu_result.json')
NCCL_RESULT = os.path.join(ROOT, 'gpurun_out', 'nccl_gpu_result.json')
BENCH2_RESULT = os.path.join(ROOT, 'gpurun_out', 'bench2_gpu_result.json')
E2E2_RESULT = os.path.join(ROOT, 'gpurun_out', 'e2e2_gpu_result.json')
# one stamp per pytest process, whichever module instance of this file asks (pytest may import it under two names)
SESSION_STAMP = os.environ.setdefault('PSN_TEST_SESSION', '%d-%d' % (os.getpid(), int(__import__('time').time())))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.hookimpl(trylast=True)
def pytest_collection_modifyitems(session, config, items):
    """The 2-rank data-parallel run of the HIP path (tests/dp_gpu_worker.py, checked by tests/test_dp_gpu.py) is a set of
    child processes.  It is started HERE -- after collection and after the -m / -k deselection (trylast), only when
    test_dp_gpu is among the SELECTED items, and before any test of this pytest process has initialised HIP: on the GPU pool
    a process that has touched the GPU must not exec another program, so the children are launched while this process is
    still GPU-free (torch.cuda.device_count() does not initialise the device on this image) and the test later only reads the
    result file.  The result carries this session's stamp; the test rejects a file left behind by another session."""
    if config.getoption('collectonly'):
        return
    jobs = [(res, worker) for key, res, worker in (('test_dp_gpu', DP_RESULT, 'dp_gpu_worker.py'), ('test_nccl_gpu', NCCL_RESULT, 'nccl_gpu_worker.py'),
                                   ('test_bench_multirank_gpu', BENCH2_RESULT, 'bench2_gpu_worker.py'), ('test_e2e_dp_gpu', E2E2_RESULT, 'e2e2_gpu_worker.py'))
            if any(key in it.nodeid for it in items)]
    if not jobs:
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    for result, worker in jobs:
        os.makedirs(os.path.dirname(result), exist_ok=True)
        for f in (result, result + '.stamp'):
            if os.path.exists(f):
                os.remove(f)
        log = open(result + '.log', 'w')
        try:
            subprocess.run([sys.executable, os.path.join(ROOT, 'tests', worker), result], env=env, stdout=log,
                           stderr=subprocess.STDOUT, timeout=900)
            with open(result + '.stamp', 'w') as f:
                f.write(SESSION_STAMP)
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
