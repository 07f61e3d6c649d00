#!/bin/bash
# round-2 first GPU pass: parity report + tests + bench (1 rank, 2 gloo ranks on one device)
mkdir -p gpurun_out/r02a
O=gpurun_out/r02a
timeout 120 python tools/dbg/probe_exec.py > $O/probe_exec.txt 2>&1
PSN_PARITY_REPORT=1 timeout 900 python -m pytest tests -q -m gpu -x -s > $O/report.txt 2>&1
echo "report rc=$?" >> $O/report.txt
timeout 900 python -m pytest tests -q -m gpu > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/tests.txt
cp gpurun_out/dp_gpu_result.json* $O/ 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench1.json 2> $O/bench1.err
echo "bench1 rc=$?" >> $O/bench1.err
timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --backend gloo --single-device > $O/bench2.json 2> $O/bench2.err
echo "bench2 rc=$?" >> $O/bench2.err
tail -5 $O/tests.txt; tail -3 $O/bench1.err; tail -3 $O/bench2.err
