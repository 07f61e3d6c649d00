#!/bin/bash
O=gpurun_out/r02d
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt
tail -4 $O/tests.txt
timeout 300 python tools/dbg/bench_chains.py > $O/bench_chains.txt 2>&1
cat $O/bench_chains.txt
