#!/bin/bash
# Per-chain hardware counters of ops.GeoFieldFused (F1 value pass, F2 sweep, B1 sweep adjoint, B2 value adjoint) at
# 262144 query points: HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and matrix-pipe occupancy.
# Writes gpurun_out/$1/pmc_chains.json (copied to profiles/ by hand).  rocprofv3 runs python3 directly (no wrapper).
TAG=${1:-r02}
MODE=${2:-chains-only}   # 'single' = the single-dump experiment (JSON: pmc_chains_single.json); 'x3' = + split-bf16 weight stages (pmc_chains_x3.json)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
    i=$((i+1))
    rm -rf /tmp/pc$i
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pc$i -o c -- python3 $R/tools/dbg/bench_chains.py 262144 $MODE > $OUT/pmc_chains_run$i.log 2>&1
    F=$(find /tmp/pc$i -name '*counter_collection*' | head -1)
    cp "$F" $OUT/pmc_chains_pass$i.csv 2>/dev/null
done
python3 - $OUT $MODE <<'PY'
import csv, json, sys, os
out = sys.argv[1]
res = {}
names = ['F1 value pass', 'F2 sweep', 'B1 sweep adjoint', 'B2 value adjoint']
for i in (1, 2, 3):
    p = os.path.join(out, 'pmc_chains_pass%d.csv' % i)
    if not os.path.exists(p):
        continue
    rows = [r for r in csv.DictReader(open(p)) if 'mlp_infer_kernel<true' in r['Kernel_Name']]
    ids = sorted(set(int(r['Dispatch_Id']) for r in rows))
    sel = ids[8:12]  # third iteration of the four chains
    for d, name in zip(sel, names):
        for r in rows:
            if int(r['Dispatch_Id']) == d:
                res.setdefault(name, {})[r['Counter_Name']] = float(r['Counter_Value'])
Q = 262144
alg = {'F1 value pass': (0, 1 if sys.argv[2] in ('single', 'x3') else 2), 'F2 sweep': (1, 2), 'B1 sweep adjoint': (2, 2), 'B2 value adjoint': (2, 1)}
for name, v in res.items():
    rd, wr = alg[name]
    v['algorithmic_read_bytes'] = rd * 8 * 1024 * Q
    v['algorithmic_write_bytes'] = wr * 8 * 1024 * Q
    if 'FETCH_SIZE' in v:
        v['fetch_bytes_corrected'] = 2 * v['FETCH_SIZE'] * 1024  # gfx950: 128-B requests tallied as 64 B (MI355X_MICROARCH.md, HBM)
    if 'WRITE_SIZE' in v:
        v['write_bytes'] = v['WRITE_SIZE'] * 1024
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'GRBM_GUI_ACTIVE' in v:
        v['mfma_busy_frac'] = round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (v['GRBM_GUI_ACTIVE'] / 8.0), 4)
json.dump({'_comment': 'rocprofv3 --pmc (3 separate passes, --kernel-trace only) on tools/dbg/bench_chains.py, 262144 query points, '
           'third iteration; FETCH_SIZE / WRITE_SIZE in KB; read side doubled per MI355X_MICROARCH.md', 'rows': Q, 'chains': res},
          open(os.path.join(out, {'single': 'pmc_chains_single.json', 'x3': 'pmc_chains_x3.json'}.get(sys.argv[2], 'pmc_chains.json')), 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
