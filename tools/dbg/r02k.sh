#!/bin/bash
O=gpurun_out/r02k
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt
tail -30 $O/tests.txt
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-stage1 --no-extra > $O/bench1.json 2> $O/bench1.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r02k/bench1.json') if l.startswith('{')][-1])
print('stage2 ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'loss', d['loss'])
PY
