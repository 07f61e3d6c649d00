"""GPU idle gaps from a rocprofv3 kernel trace: python gaps.py kernel_trace.csv [min_gap_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]) for r in rows))
t0 = ev[0][0]
busy_end = ev[0][1]
tot_gap = 0
for i in range(1, len(ev)):
    s, e, n = ev[i]
    gap = (s - busy_end) / 1e3
    if gap > thr:
        print('gap %8.1f us at %9.2f ms  after [%s]  before [%s]' % (gap, (s - t0) / 1e6, ev[i - 1][2], n))
    if gap > 0:
        tot_gap += gap
    busy_end = max(busy_end, e)
print('span %.2f ms, total idle %.2f ms' % ((busy_end - t0) / 1e6, tot_gap / 1e3))
