#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r02l
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o s2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-stage1 --no-extra > $O/s2.json 2> $O/s2.err
F=$(find /tmp/p2 -name '*kernel_stats.csv' | head -1)
cp $F $O/s2_kernel_stats.csv
python3 - $F <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=12
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel ms/step %.3f  launches/step %.1f' % (tot/steps/1e6, sum(int(r['Calls']) for r in rows)/steps))
big=[r for r in rows if float(r['AverageNs'])>20000]
small=[r for r in rows if float(r['AverageNs'])<=20000]
print('small kernels: %.3f ms/step in %.1f launches' % (sum(float(r['TotalDurationNs']) for r in small)/steps/1e6, sum(int(r['Calls']) for r in small)/steps))
for r in big: print('%-90s %6.1f/step %9.1f us/step' % (r['Name'][:90], int(r['Calls'])/steps, float(r['TotalDurationNs'])/steps/1e3))
PY
tail -1 $O/s2.json | cut -c1-200
