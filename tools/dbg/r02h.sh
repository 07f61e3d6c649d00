#!/bin/bash
O=gpurun_out/r02h
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt
tail -5 $O/tests.txt
timeout 600 python tools/bench_shadow.py > $O/shadow.json 2> $O/shadow.err; cat $O/shadow.json; tail -3 $O/shadow.err
