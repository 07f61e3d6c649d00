"""Timeline of the LAST replayed step in a rocprofv3 kernel trace CSV: per kernel start offset, duration, gap to the previous end."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step starts at the marker kernel (default: light_rows_fwd, the first launch of a stage-2 step)
marker = sys.argv[2] if len(sys.argv) > 2 else 'light_rows_fwd'
starts = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
a, b = starts[-3], starts[-2]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
print('kernels in step', len(step), 'span %.1f us' % ((max(int(r['End_Timestamp']) for r in step) - t0) / 1e3))
busy_end = t0
idle = 0.0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - busy_end) / 1e3
    if gap > 0: idle += gap
    print('%8.1f %8.1f %6.1f  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, r.get('Queue_Id', '?'), r['Kernel_Name'][:90]))
    busy_end = max(busy_end, e)
print('GPU idle inside the step: %.1f us' % idle)
