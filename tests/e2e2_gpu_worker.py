#!/usr/bin/env python3
"""BASELINE configs[4] as one chain with TWO data-parallel ranks on ONE GPU: `torch.distributed.run --nproc-per-node 2 tools/run_e2e.py
--backend gloo --single-device --graph ...` -- stage 1 ray-parallel, stage 2 pixel-parallel with every rank gathering only its slice
from the device-resident view store (handoff.DeviceViews(dp=...)), the step replayed from HIP graphs around the collectives.  Guards
the training LOOP the 8-GPU configuration runs against hangs and rank divergence; compared by the test with the single-rank run of
the same seed.  Started by tests/conftest.py before pytest initialises HIP.    python tests/e2e2_gpu_worker.py OUT.json"""
import json
import os
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ['--h', '40', '--w', '40', '--views', '2', '--lights', '5', '--light-bs', '3', '--vis-plus', '6', '--vis-train-num', '3', '--rays', '192',
        '--s1-steps', '16', '--s2-steps', '32', '--pixels', '600', '--envmap-h', '4', '--graph']


def run(cmd, env=None):
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400, env=env)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr.decode()[-3000:]


def main():
    out_path = sys.argv[1]
    res = {'ok': False}
    try:
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        d2, d1 = tempfile.mkdtemp(prefix='e2e2_'), tempfile.mkdtemp(prefix='e2e1_')
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        rc2, line2, err2 = run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                                '--master-port', str(port), os.path.join(ROOT, 'tools', 'run_e2e.py')] + ARGS +
                               ['--backend', 'gloo', '--single-device', '--out', os.path.join(d2, 'shape')], env)
        rc1, line1, err1 = run([sys.executable, os.path.join(ROOT, 'tools', 'run_e2e.py')] + ARGS + ['--out', os.path.join(d1, 'shape')], env)
        res.update(rc2=rc2, rc1=rc1, two_ranks=line2, one_rank=line1)
        if rc2 != 0 or line2 is None:
            res['stderr2'] = err2
        if rc1 != 0 or line1 is None:
            res['stderr1'] = err1
        res['ok'] = rc2 == 0 and rc1 == 0 and line2 is not None and line1 is not None
    except subprocess.TimeoutExpired as e:
        res['error'] = 'timeout: %s' % e
    with open(out_path, 'w') as f:
        json.dump(res, f, indent=1)
    sys.exit(0 if res['ok'] else 1)


if __name__ == '__main__':
    main()
