#!/bin/bash
O=gpurun_out/r02m
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "composite or stage1 or unisurf or march" > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt
tail -4 $O/tests.txt
timeout 300 python tools/bench_composite.py > $O/composite.json 2>$O/composite.err
python -c "
import json; d=json.load(open('$O/composite.json'))
for k,v in d['kernels'].items(): print(k, round(v['ms'],3),'ms', round(v['achieved_GBps']),'GB/s', round(v['frac_of_peak'],3))"
