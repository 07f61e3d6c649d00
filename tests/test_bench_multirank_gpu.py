"""bench.py's multi-rank path (what the driver runs at N = 2, 4, 8) with two ranks on the one GPU of the test box: completes without
hanging, prints one JSON line with the contract's keys, and refuses more GPUs than the node has with exit code 2."""
import json
import os

import pytest

from tests.conftest import BENCH2_RESULT, SESSION_STAMP

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_on_one_gpu(cuda):
    assert os.path.exists(BENCH2_RESULT), 'the bench worker left no result: %s' % (
        open(BENCH2_RESULT + '.log').read()[-3000:] if os.path.exists(BENCH2_RESULT + '.log') else 'no log')
    stamp = open(BENCH2_RESULT + '.stamp').read() if os.path.exists(BENCH2_RESULT + '.stamp') else None
    assert stamp == SESSION_STAMP, 'stale result (session %r, this is %r): run with -m gpu' % (stamp, SESSION_STAMP)
    res = json.load(open(BENCH2_RESULT))
    assert res['ok'], json.dumps(res, indent=1)[:4000]
    line = res['line']
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in line, k
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0 and line['allreduce_bytes'] > 2_000_000
    assert line['config']['parallelism'] == 'pixel-dp2' and line['roofline']['frac'] <= 1.0
    # BASELINE cfg 4 on the line (32768 px split over the ranks), eagerly AND replayed from HIP graphs around the collectives
    c4 = line['strong_cfg4']
    assert 'error' not in c4, c4
    assert c4['pixels_per_gpu'] == 16384 and c4['eager']['ms_per_step'] > 0 and c4['graph']['ms_per_step'] > 0, c4
    assert res['too_many_gpus_rc'] == 2 and 'GPU(s) are visible' in res['too_many_gpus_msg']
