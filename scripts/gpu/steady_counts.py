"""Kernel launches per training step in the steady state, by queue, from a rocprofv3 --kernel-trace CSV of
`bench.py --steps S ...`: the trace is cut at the S last optimizer launches (multi_tensor_apply ... FusedAdam), the
dispatches between the first and the last of them are counted and divided by S - 1.
usage: steady_counts.py KERNEL_TRACE_CSV [top]"""
import csv
import sys
from collections import Counter, defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'FusedAdam' in r['Kernel_Name']]
# one optimizer step = a burst of Adam launches: bursts are separated by > 1 ms
bursts = []
for i in adam:
    t = int(rows[i]['Start_Timestamp'])
    if not bursts or t - bursts[-1][1] > 1_000_000:
        bursts.append([i, t])
    bursts[-1][1] = t
# the last two bursts belong to bench.py's host_calls() (a second model: its construction and optimizer-state fills
# are not part of a step): the window ends before them
SKIP = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n_steps = min(len(bursts) - 1 - SKIP, 10)
lo, hi = bursts[len(bursts) - 1 - SKIP - n_steps][0], bursts[len(bursts) - 1 - SKIP][0]
seg = rows[lo:hi]
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6 / n_steps
by_q = defaultdict(Counter)
dur = defaultdict(float)
for r in seg:
    q = r.get('Queue_Id', '?')
    name = r['Kernel_Name']
    by_q[q][name] += 1
    dur[(q, name)] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('steady state: %d steps, %.3f ms wall per step' % (n_steps, span))
for q, c in sorted(by_q.items(), key=lambda kv: -sum(kv[1].values())):
    tot = sum(c.values())
    busy = sum(v for (qq, _), v in dur.items() if qq == q) / 1e3 / n_steps
    lib = sum(v for k, v in c.items() if not (k.startswith('void at::') or k.startswith('__amd_rocclr') or 'at::native' in k))
    print('queue %s: %.1f launches per step (%.1f of the library, %.1f torch / runtime), %.3f ms of kernels per step'
          % (q, tot / n_steps, lib / n_steps, (tot - lib) / n_steps, busy))
    for name, n in c.most_common(top):
        print('   %7.1f /step  %8.1f us avg   %s' % (n / n_steps, dur[(q, name)] / n, name[:110]))
