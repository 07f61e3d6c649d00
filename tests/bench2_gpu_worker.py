#!/usr/bin/env python3
"""The multi-rank path of bench.py on ONE GPU: `python bench.py --gpus 2 --backend gloo --single-device` (the self-launching
parent starts two rank processes through torch.distributed.run; both use cuda:0, gloo carries the collectives).  Guards what the
driver's N = 2 / 4 / 8 runs execute -- sharding, the barrier-bracketed timed region, the instrumented pass and the launch count on
every rank, rank 0's JSON line -- against hangs: the whole run is under a hard timeout.  Started by tests/conftest.py before pytest
initialises HIP; never touches the GPU itself.    python tests/bench2_gpu_worker.py OUT.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out_path = sys.argv[1]
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--backend', 'gloo', '--single-device',
           '--no-stage1', '--no-cpu-baseline', '--pixels', '4096', '--cfg4-graph']   # (--cfg4-graph: the N > 1 graph capture of the diagnostic is opt-in)
    res = {'ok': False, 'cmd': ' '.join(cmd)}
    try:
        p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
        lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
        res['rc'] = p.returncode
        if p.returncode == 0 and lines:
            res['line'] = json.loads(lines[-1])
            res['ok'] = True
        else:
            res['stderr'] = p.stderr.decode()[-3000:]
        q = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '64'], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        res['too_many_gpus_rc'], res['too_many_gpus_msg'] = q.returncode, q.stderr.decode()[-300:]
    except subprocess.TimeoutExpired as e:
        res['error'] = 'timeout: %s' % e
    with open(out_path, 'w') as f:
        json.dump(res, f, indent=1)
    sys.exit(0 if res['ok'] else 1)


if __name__ == '__main__':
    main()
