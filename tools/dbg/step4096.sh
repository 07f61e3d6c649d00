#!/bin/bash
# 4096-px replayed rank shard: wall / gpu ms (strong_projection) + its kernel timeline.  usage: tools/dbg/step4096.sh <out dir under gpurun_out>
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/strong_projection.py --pixels 32768 4096 --graph --queue-ahead --steps 40 --no-profiler --out $O/strong.json > $O/strong.log 2>&1
rm -rf /tmp/tl_x
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_x -o t -- python3 $R/tools/dbg/trace_step.py 4096 graph > /dev/null 2>&1
python3 $R/tools/dbg/trace_step_analyse.py $(find /tmp/tl_x -name '*kernel_trace.csv' | head -1) > $O/timeline_4096_graph.txt 2>&1
python3 - <<PY
import json
j = json.load(open('$O/strong.json'))
for c in j['cases']:
    print(c['pixels'], {k: (c[k]['wall_ms'], c[k].get('gpu_ms')) for k in ('eager', 'graph') if k in c})
PY
head -1 $O/timeline_4096_graph.txt; tail -1 $O/timeline_4096_graph.txt
