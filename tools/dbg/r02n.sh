#!/bin/bash
O=gpurun_out/r02n
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "composite or stage1" > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt
tail -4 $O/tests.txt
for S in 128 96; do
timeout 300 python tools/bench_composite.py --iters 30 --samples $S > $O/c$S.json 2>$O/c$S.err
python -c "
import json; d=json.load(open('$O/c$S.json'))
print('S=$S', ' '.join('%s %.0f' % (k, v['achieved_GBps']) for k,v in d['kernels'].items()))"
done
