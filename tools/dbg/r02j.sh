#!/bin/bash
for i in 1 2; do
python tools/bench_stage1.py --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused  ', round(d['ms_per_step'],2))"
python tools/bench_stage1.py --steps 10 --warmup 3 --compact-secant 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('compact', round(d['ms_per_step'],2))"
done
nproc; python -c "import os;print(os.sched_getaffinity(0).__len__())"
