#!/bin/bash
# SQ counter passes over the bf16 inference kernel alone (tools/dbg/run_bf16_only.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
         "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
    i=$((i+1))
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pb$i -o c -- python3 $R/tools/dbg/run_bf16_only.py > /dev/null 2>&1
    F=$(find /tmp/pb$i -name '*counter_collection*' | head -1)
    python3 - "$F" <<'PY'
import csv, sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    if 'mlp_infer_bf16_kernel' in r['Kernel_Name']:
        acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, v in acc.items():
    print('%-28s %.4e (last of %d dispatches)' % (k, v[-1], len(v)))
PY
done
