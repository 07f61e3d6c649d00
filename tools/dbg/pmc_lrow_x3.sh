#!/bin/bash
# matrix-pipe occupancy and effective clock of the shading-row launch, fp32 vs split-bf16 weight stages (own PMC pass, --kernel-trace only)
# usage: tools/dbg/pmc_lrow_x3.sh <tag>
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_lx3 /tmp/kt_lx3
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_lx3 -o c -- python3 $R/tools/dbg/bench_lrow_x3.py > /dev/null 2>&1
F=$(find /tmp/pmc_lx3 -name '*counter_collection*' | head -1)
(head -1 $F; grep "mlp_infer_kernel" $F | head -60) > $O/pmc_lrow_x3.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_lx3 -o k -- python3 $R/tools/dbg/bench_lrow_x3.py > /dev/null 2>&1
cp $(find /tmp/kt_lx3 -name '*kernel_stats*' | head -1) $O/lrow_x3_kernel_stats.csv
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open('$O/pmc_lrow_x3.csv')))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and 'GRBM_GUI_ACTIVE' in m:
        print(k, 'mfma busy frac %.3f' % (m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (m['GRBM_GUI_ACTIVE'] / 8.0)), 'gui cycles per XCD %.3e' % (m['GRBM_GUI_ACTIVE'] / 8.0), {n: '%.3e' % v for n, v in m.items()})
PY
head -4 $O/lrow_x3_kernel_stats.csv | cut -c1-200
