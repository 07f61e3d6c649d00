#!/bin/bash
O=gpurun_out/r02c
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt
tail -15 $O/tests.txt
timeout 300 python tools/dbg/bench_chains.py > $O/bench_chains.txt 2>&1
cat $O/bench_chains.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench1.json 2> $O/bench1.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r02c/bench1.json') if l.startswith('{')][-1])
print('stage2 ms', d['ms_per_step'], 'roofline', d['roofline']['frac'])
print(json.dumps(d['stage1'], indent=1))
PY
