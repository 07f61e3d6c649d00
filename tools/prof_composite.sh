#!/bin/bash
# rocprofv3 record of the alpha-composite kernels at a size where they are HBM-bound (2 M rays x 128 samples, tools/bench_composite.py):
# --kernel-trace --stats for the average durations, and one --pmc pass each for FETCH_SIZE and WRITE_SIZE (--kernel-trace only).
# usage: tools/prof_composite.sh <tag>   -> gpurun_out/<tag>/composite_{kernel_stats.csv,pmc.json,bench.json}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python3 $R/tools/bench_composite.py 2>/dev/null | tail -1 > $O/composite_bench.json
rm -rf /tmp/pc_kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_kt -o k -- python3 $R/tools/bench_composite.py > /dev/null 2>&1
cp $(find /tmp/pc_kt -name '*kernel_stats*' | head -1) $O/composite_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pc_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pc_$C -o c -- python3 $R/tools/bench_composite.py --iters 2 > /dev/null 2>&1
  F=$(find /tmp/pc_$C -name '*counter_collection*' | head -1)
  (head -1 $F; grep composite_ $F) > $O/composite_pmc_$C.csv
done
python3 - "$O" <<'PY'
import csv, sys, os, json, collections
O = sys.argv[1]
N, S = 2 * 1024 * 1024, 128
algo = {'composite_fwd_kernel': (20 * S + 16) * N, 'composite_fwd_flat_kernel': (16 * S + 16) * N, 'composite_bwd_kernel': (36 * S + 16) * N,
        'composite_acc4_kernel': (4 * S + 4) * N, 'composite_fwd_multi_kernel': (20 * S + 16) * N}
res = {}
for r in csv.DictReader(open(os.path.join(O, 'composite_kernel_stats.csv'))):
    for k, b in algo.items():
        if k + '<' in r['Name'] or r['Name'].startswith('void psn::' + k) or ('psn::' + k) in r['Name']:
            avg_ns = float(r['AverageNs'])
            res[k] = {'calls': int(r['Calls']), 'avg_us': round(avg_ns / 1e3, 2), 'min_us': round(float(r['MinNs']) / 1e3, 2), 'algorithmic_bytes': b,
                      'achieved_TBps_rocprof_avg': round(b / avg_ns * 1e-3, 3), 'achieved_TBps_rocprof_min': round(b / float(r['MinNs']) * 1e-3, 3)}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(O, 'composite_pmc_%s.csv' % c))):
        for k in algo:
            if ('psn::' + k) in r['Kernel_Name']:
                acc[k].append(float(r['Counter_Value']))
    for k, v in acc.items():
        res.setdefault(k, {})[c + '_KB_last_dispatch'] = v[-1]
for k, d in res.items():
    if 'FETCH_SIZE_KB_last_dispatch' in d and 'WRITE_SIZE_KB_last_dispatch' in d:
        # MI355X_MICROARCH.md (HBM): gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled; WRITE_SIZE as reported
        d['counter_bytes'] = int((2 * d['FETCH_SIZE_KB_last_dispatch'] + d['WRITE_SIZE_KB_last_dispatch']) * 1024)
        d['counter_over_algorithmic'] = round(d['counter_bytes'] / d['algorithmic_bytes'], 3)
json.dump({'rays': N, 'samples': S, 'kernels': res}, open(os.path.join(O, 'composite_pmc.json'), 'w'), indent=1)
print(json.dumps(res))
PY
