#!/bin/bash
# HBM/fabric traffic of the visibility launch per workgroup order: one rocprofv3 --pmc pass per counter and order (--kernel-trace only).
# usage: tools/pmc_block_order.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for ORDER in row point; do
  for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    N=$(echo $C | cut -d' ' -f1)
    rm -rf /tmp/pbo_${ORDER}_$N
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pbo_${ORDER}_$N -o c -- python3 $R/tools/ab_block_order.py --order $ORDER --iters 2 > /dev/null 2>&1
    F=$(find /tmp/pbo_${ORDER}_$N -name '*counter_collection*' | head -1)
    (head -1 $F; grep mlp_infer_kernel $F) > $O/pmc_block_order_${ORDER}_$N.csv
  done
done
python3 - "$O" <<'PY'
import csv, sys, json, collections, os
O = sys.argv[1]
res = {}
for order in ('row', 'point'):
    for n in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES'):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(os.path.join(O, 'pmc_block_order_%s_%s.csv' % (order, n)))):
            k = 'bf16x3' if 'true>' in r['Kernel_Name'].split('(')[0][-8:] else 'fp32'
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, c in acc.items():
            for cn, v in c.items():
                res.setdefault(order, {}).setdefault(k, {})[cn] = v[-1]
for order, d in res.items():
    for k, c in d.items():
        # FETCH_SIZE / WRITE_SIZE: the guide's gfx950 correction -- FETCH_SIZE counts 32 B units x 2 (64 B requests), WRITE_SIZE in KB
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            c['traffic_GB'] = round((2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / 1e9, 3)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
            c['mfma_busy_frac'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (c['GRBM_GUI_ACTIVE'] / 8.0), 4)
json.dump(res, open(os.path.join(O, 'pmc_block_order.json'), 'w'), indent=1)
print(json.dumps(res))
PY
