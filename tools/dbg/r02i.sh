#!/bin/bash
O=gpurun_out/r02i
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt
tail -3 $O/tests.txt
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r02i/bench1.json') if l.startswith('{')][-1])
print('stage2 ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'])
s=d['stage1']; print('stage1 ms', s['ms_per_step'], {k:(v.get('ms_per_step'), v.get('achieved')) for k,v in s.items() if isinstance(v,dict)})
PY
