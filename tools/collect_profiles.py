#!/usr/bin/env python3
"""Reduce gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into the tracked profiles/ files:
kernel-stats CSVs, the bench JSON lines and profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

    python tools/collect_profiles.py r01h
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_rows(path):
    out = {}
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        key = (r['Dispatch_Id'], r['Kernel_Name'])
        out.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
    return out


def byte_model(ns, L, V):
    """Byte counts of the dominant launch ((L + V) * ns rows of the fused visibility MLP), stated separately:
    compulsory  what must cross HBM if every table / weight is read once: U [ns, 512] and V [L + V, 512] init tables, ~2 MB of
                packed weights, 4 B of output per row, and the 9 x 1 KB activation dumps of the V * ns supervised rows;
    l2_request  what the waves REQUEST: every row reads its 2 KB U row and its 2 KB V row (L2 / Infinity-Cache hits for all
                but the first use) + output + dumps.  rocprofv3's FETCH_SIZE counts fabric-side requests incl. Infinity-Cache
                hits, so the measured figure lies between the two."""
    rows = (L + V) * ns
    dumps = V * ns * 9 * 1024
    return {'compulsory_bytes_per_launch': int(ns * 2048 + (L + V) * 2048 + 2 * 2 ** 20 + rows * 4 + dumps),
            'compulsory_bytes_note': 'U [Ns, 512] + V [L+V, 512] fp32 tables and ~2 MB of packed weights read once, 4 B out per '
                                     'row, 9 x 1 KB activation dumps for the V*Ns supervised rows (DESIGN.md 5)',
            'l2_request_bytes_per_launch': int(rows * (4096 + 4) + dumps),
            'l2_request_bytes_note': 'per row 2 x 2 KB init-table rows as requested by the waves (cache hits for all but the '
                                     'first use) + 4 B out; + the dumps'}


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, 'gpurun_out', tag)
    dst = os.path.join(ROOT, 'profiles')
    for name in ('bench_stage2.json', 'bench_stage1.json', 'bench_stage2_kernel_stats.csv', 'bench_stage1_kernel_stats.csv'):
        p = os.path.join(src, name)
        if os.path.exists(p):
            if name.endswith('.json'):  # keep the JSON line only
                lines = [l for l in open(p).read().splitlines() if l.startswith('{')]
                open(os.path.join(dst, '%s_%s' % (tag, name)), 'w').write(lines[-1] + '\n')
            else:
                shutil.copy(p, os.path.join(dst, '%s_%s' % (tag, name)))
    for name in ('march_sweep.json', 'bf16x6_kernel.json', 'bf16x6_kernel_zero_operands.json', 'tn256.txt', 'shadow_visibility.json',
                 'composite.json', 'pmc_x3.csv', 'strong_projection.json', 'x3occ.txt', 'strong4096_kernel_stats.csv',
                 'timeline_4096_graph.txt', 'timeline_32768.txt', 'stage1_rank_shards.jsonl', 'e2e_full.json', 'e2e_full_bf16x3.json', 'tn256_x3.json', 'ab_single_dump.json', 'ab_chain_x3.json', 'lrow_x3.json',
                 'ab_block_order.json', 'pmc_block_order.json', 'composite_kernel_stats.csv', 'composite_pmc.json', 'composite_bench.json',
                 'pmc_chains_single.json'):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(dst, '%s_%s' % (tag, name)))
    bench = json.loads(open(os.path.join(dst, '%s_bench_stage2.json' % tag)).read())
    fetch = counter_rows(os.path.join(src, 'pmc_FETCH_SIZE.csv'))
    write = counter_rows(os.path.join(src, 'pmc_WRITE_SIZE.csv'))
    busy = counter_rows(os.path.join(src, 'pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv'))

    def dominant(rows, counter):  # the forward launch = the largest value among the lean-kernel dispatches
        vals = [v[counter] for (d, k), v in rows.items() if 'mlp_infer_kernel<false' in k and counter in v]
        return max(vals) if vals else None

    f_kb, w_kb = dominant(fetch, 'FETCH_SIZE'), dominant(write, 'WRITE_SIZE')
    res = {
        '_comment': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (separate '
                    'passes, --kernel-trace only) on `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`; values of the '
                    'dominant dispatch (mlp_infer_kernel<false>, (L+V)*Ns rows).  Units: the counters report KB -> x 1024.  Per '
                    'MI355X_MICROARCH.md (HBM section) gfx950 FETCH_SIZE tallies 128-B fabric requests as 64 B for wide '
                    '16 B/lane reads, so the read side is doubled; WRITE_SIZE is taken as reported (uncalibrated); '
                    'Infinity-Cache hits are included in both.',
        'kernel': 'psn::mlp_infer_kernel<false, 16>',
        'rows_per_launch': bench['config']['surface_pixels_total'] * (bench['config']['lights'] + bench['config']['vis_lights']),
        'FETCH_SIZE_KB_raw': f_kb, 'WRITE_SIZE_KB_raw': w_kb,
    }
    if f_kb is not None and w_kb is not None:
        res['hbm_side_bytes_per_launch'] = int(2 * f_kb * 1024 + w_kb * 1024)
    ns, L, V = bench['config']['surface_pixels_total'], bench['config']['lights'], bench['config']['vis_lights']
    res.update(byte_model(ns, L, V))
    if res.get('hbm_side_bytes_per_launch'):
        res['traffic_over_compulsory'] = round(res['hbm_side_bytes_per_launch'] / res['compulsory_bytes_per_launch'], 2)
    mb = [(v.get('SQ_VALU_MFMA_BUSY_CYCLES'), v.get('GRBM_GUI_ACTIVE')) for (d, k), v in busy.items() if 'mlp_infer_kernel<false' in k]
    mb = [x for x in mb if x[0] and x[1]]
    if mb:
        b, g = max(mb, key=lambda x: x[0])
        res['SQ_VALU_MFMA_BUSY_CYCLES'], res['GRBM_GUI_ACTIVE'] = b, g
        # SQ counter: summed over 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE: summed over the 8 XCDs
        res['mfma_busy_frac'] = round(b / 1024.0 / (g / 8.0), 4)
        res['mfma_busy_note'] = 'SQ_VALU_MFMA_BUSY_CYCLES / (256 CUs x 4 SIMDs) over GRBM_GUI_ACTIVE / 8 XCDs'
    json.dump(res, open(os.path.join(dst, 'pmc_traffic.json'), 'w'), indent=2)
    print(json.dumps(res, indent=2))


if __name__ == '__main__':
    main()
