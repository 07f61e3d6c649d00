"""Data parallelism of the HIP product path with world_size 2 on one GPU: the rank processes are started by
tests/conftest.py::pytest_sessionstart (before this process initialises HIP) and run tests/dp_gpu_worker.py; this test
checks their verdict: summed rank gradients == single-rank HIP gradients for the stage-2 TrainStep (dense weights and
the light tables, ragged 501 / 500 pixel shards), the stage-1 Trainer (65 / 64 rays), and three steps of the training loop with every
rank gathering only ITS pixel slice from the device-resident view store (handoff.DeviceViews(dp=...).loader: 301 / 300 pixels)."""
import json
import os

import pytest

from tests.conftest import DP_RESULT, SESSION_STAMP

pytestmark = pytest.mark.gpu


def test_two_rank_hip_gradients_match_single_rank(cuda):
    assert os.path.exists(DP_RESULT), 'the 2-rank worker left no result: %s' % (
        open(DP_RESULT + '.log').read()[-3000:] if os.path.exists(DP_RESULT + '.log') else 'no log')
    stamp = open(DP_RESULT + '.stamp').read() if os.path.exists(DP_RESULT + '.stamp') else None
    assert stamp == SESSION_STAMP, 'stale 2-rank result (written by session %r, this is %r): run with -m gpu' % (stamp, SESSION_STAMP)
    res = json.load(open(DP_RESULT))
    assert res['ok'], json.dumps(res, indent=1)[:4000]
    assert res['n_checks'] > 100
    assert any(k.startswith('device views') for _, k in res['worst']) or res['n_checks'] > 280   # the DeviceViews loop ran (~70 more checks)
