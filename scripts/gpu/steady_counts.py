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

# ---- idle time of the main queue: gaps between consecutive dispatches of the busiest queue, by size and by the kernel
#      that follows the gap (a join with a side stream, a host-bound stretch and the plain dispatch gap look different)
main_q = max(by_q.items(), key=lambda kv: sum(kv[1].values()))[0]
mq = [r for r in seg if r.get('Queue_Id', '?') == main_q]
gaps = []
for a, b in zip(mq[:-1], mq[1:]):
    g = (int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3
    gaps.append((g, a['Kernel_Name'][:60], b['Kernel_Name'][:60]))
tot_gap = sum(max(g, 0.0) for g, _, _ in gaps) / n_steps
print('queue %s gaps: %.3f ms per step between consecutive dispatches' % (main_q, tot_gap / 1e3))
edges = [0, 2, 4, 6, 8, 12, 20, 50, 100, 1e9]
for lo_e, hi_e in zip(edges[:-1], edges[1:]):
    sel = [g for g, _, _ in gaps if lo_e <= g < hi_e]
    print('   gaps of %4g..%-6g us: %7.1f per step, %8.1f us per step' % (lo_e, hi_e, len(sel) / n_steps, sum(sel) / n_steps))
after = defaultdict(lambda: [0, 0.0])
for g, a, b in gaps:
    if g >= 8:
        after[(a, b)][0] += 1
        after[(a, b)][1] += g
print('   gaps >= 8 us, by (kernel before -> kernel after):')
for (a, b), (n, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:25]:
    print('   %6.1f /step %8.1f us avg   %s -> %s' % (n / n_steps, t / n, a, b))

# ---- what the other queues run while the main queue waits: for the gaps in front of the optimizer (the join at the end
#      of the backward pass) the dispatches of the side queues that overlap the gap, in start order (last step of the window)
big = [(int(a['End_Timestamp']), int(b['Start_Timestamp']), b['Kernel_Name'][-60:]) for a, b in zip(mq[:-1], mq[1:])
       if int(b['Start_Timestamp']) - int(a['End_Timestamp']) > 300_000 and
       ('multi_tensor_apply' in b['Kernel_Name'] or 'conv_lean' in b['Kernel_Name'])]
for t0, t1, nm in big[-2:]:
    print('gap of %.1f us in front of %s: the side queues meanwhile' % ((t1 - t0) / 1e3, nm))
    for r in seg:
        if r.get('Queue_Id', '?') == main_q:
            continue
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if e > t0 and s < t1:
            print('   queue %s  start %+8.1f us  %7.1f us   %s' % (r.get('Queue_Id', '?'), (s - t0) / 1e3, (e - s) / 1e3,
                                                                   r['Kernel_Name'][:90]))
