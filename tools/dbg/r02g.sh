#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02g
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/p1 -o s1 -- python3 $GRAFT_REPO_ROOT/tools/bench_stage1.py --steps 4 --warmup 2 > $O/s1.json 2> $O/s1.err
F=$(find /tmp/p1 -name '*kernel_trace.csv' | head -1)
python3 $GRAFT_REPO_ROOT/tools/dbg/gaps.py $F 300 | tail -40
cat $O/s1.json
