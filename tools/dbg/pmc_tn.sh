#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
         "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pt$i -o c -- python3 $R/tools/dbg/bench_tn.py > /dev/null 2>&1
    F=$(find /tmp/pt$i -name '*counter_collection*' | head -1)
    python3 - "$F" <<'PY'
import csv, sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    if 'tn256' in r['Kernel_Name']:
        acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, v in acc.items():
    print('%-28s %.4e (last of %d dispatches)' % (k, v[-1], len(v)))
PY
done
