#!/bin/bash
O=gpurun_out/r02b
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt
tail -15 $O/tests.txt
timeout 300 tools/dbg/bin/membench 262144 > $O/membench.txt 2>&1
cat $O/membench.txt
timeout 300 python tools/dbg/bench_chains.py > $O/bench_chains.txt 2>&1
cat $O/bench_chains.txt
timeout 900 bash tools/dbg/pmc_chains.sh r02b > $O/pmc_chains_stdout.txt 2>&1
tail -60 $O/pmc_chains_stdout.txt
timeout 300 python tools/run_e2e.py > $O/e2e.txt 2>&1; tail -3 $O/e2e.txt
