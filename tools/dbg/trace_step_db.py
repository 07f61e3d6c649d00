"""Timeline of one steady-state step out of a rocprofv3 rocpd database (kernels view): per kernel start offset, duration, gap.
usage: trace_step_db.py DB [marker=adam_flat] [k=-8] [--agg]"""
import sqlite3, sys, re, collections
args = [a for a in sys.argv[1:] if not a.startswith('--')]
db = sqlite3.connect(args[0])
rows = db.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()
marker = args[1] if len(args) > 1 else 'adam_flat'
idx = [i for i, r in enumerate(rows) if marker in r[0]]
print('kernels', len(rows), 'markers', len(idx))
k = int(args[2]) if len(args) > 2 else -8
a, b = idx[k] + 1, idx[k + 1] + 1
step = rows[a:b]
t0 = step[0][1]
prev_end = step[0][1]
busy = 0
iv = sorted((r[1], r[2]) for r in step)
cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
def short(n):
    n = re.sub(r'\(.*', '', n)
    n = re.sub(r'void ', '', n)
    return n[-70:]
agg = collections.OrderedDict()
for r in step:
    gap = r[1] - prev_end
    if '--agg' not in sys.argv:
        print('%9.1f us  dur %8.1f  gap %7.1f  q%-3s %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap / 1e3, r[4], short(r[0])))
    d = agg.setdefault(short(r[0]), [0, 0.0, 0.0])
    d[0] += 1; d[1] += (r[2] - r[1]) / 1e3; d[2] += max(gap, 0) / 1e3
    prev_end = max(prev_end, r[2])
if '--agg' in sys.argv:
    for n, (c, d, g) in sorted(agg.items(), key=lambda x: -x[1][1]):
        print('%4d x  dur %8.1f us  gaps-before %7.1f us  %s' % (c, d, g, n))
span = max(r[2] for r in step) - t0
print('step span %.1f us, busy (union) %.1f us, idle %.1f us, kernels %d, sum dur %.1f us' % (span / 1e3, busy / 1e3, (span - busy) / 1e3, len(step), sum(r[2] - r[1] for r in step) / 1e3))
per = [(rows[idx[i + 1]][1] - rows[idx[i]][1]) / 1e3 for i in range(max(0, len(idx) + k - 6), min(len(idx) - 1, len(idx) + k + 4))]
print('periods us around', [round(p) for p in per])
