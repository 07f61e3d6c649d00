"""RCCL smoke test on the single GPU of the test box: tests/nccl_gpu_worker.py runs in a fresh
process started by tests/conftest.py -- init_process_group('nccl', world_size=1), one stage-2 TrainStep and two stage-1 Trainer
steps through DataParallel(force=True) with every collective executed by RCCL on the current HIP stream -- and must reproduce
the non-data-parallel steps bit for bit; round 4: seven data-parallel steps replayed from HIP graphs around the RCCL collectives
(psnerf_amd/stage2/graph.py) against the eager data-parallel steps, bit for bit."""
import json
import os

import pytest

from tests.conftest import NCCL_RESULT, SESSION_STAMP

pytestmark = pytest.mark.gpu


def test_rccl_world_size_one_matches_the_plain_step(cuda):
    assert os.path.exists(NCCL_RESULT), 'the nccl worker left no result: %s' % (
        open(NCCL_RESULT + '.log').read()[-3000:] if os.path.exists(NCCL_RESULT + '.log') else 'no log')
    stamp = open(NCCL_RESULT + '.stamp').read() if os.path.exists(NCCL_RESULT + '.stamp') else None
    assert stamp == SESSION_STAMP, 'stale nccl result (session %r, this is %r): run with -m gpu' % (stamp, SESSION_STAMP)
    res = json.load(open(NCCL_RESULT))
    assert res['ok'], json.dumps(res, indent=1)[:4000]
    assert res['backend'] == 'nccl' and res['stage2_bucket_bytes'] > 2_000_000 and len(res['checks']) == 8
