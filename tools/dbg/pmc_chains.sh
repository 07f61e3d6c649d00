#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR" \
         "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM"; do
    i=$((i+1))
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pc$i -o c -- python3 $R/tools/dbg/bench_chains.py > /dev/null 2>&1
    F=$(find /tmp/pc$i -name '*counter_collection*' | head -1)
    python3 - "$F" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'mlp_infer_kernel<true>' in r['Kernel_Name']]
ids = sorted(set(int(r['Dispatch_Id']) for r in rows))
# dispatches 8..11 = the third (timed) iteration of the four chains F1, F2, B1, B2
sel = ids[8:12]
for d, name in zip(sel, ['F1', 'F2', 'B1', 'B2']):
    vals = {r['Counter_Name']: float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id']) == d}
    print(name, ' '.join('%s=%.3e' % (k.replace('SQ_', ''), v) for k, v in sorted(vals.items())))
PY
done
