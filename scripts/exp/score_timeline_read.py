"""Per-queue busy time, launches and idle gaps of the LAST marked pass of scripts/exp/score_timeline.py.
usage: score_timeline_read.py KERNEL_TRACE_CSV frames"""
import csv
import sys
from collections import Counter, defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
frames = int(sys.argv[2])
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'transpose_f32_kernel' in r['Kernel_Name']]
lo, hi = marks[-2], marks[-1]
seg = rows[lo + 1:hi]
t0, t1 = int(rows[lo]['End_Timestamp']), int(rows[hi]['Start_Timestamp'])
print('marked pass: %.3f ms per frame on the device, %d launches per frame' % ((t1 - t0) / 1e6 / frames, len(seg) / frames))
by_q = defaultdict(list)
for r in seg:
    by_q[r.get('Queue_Id', '?')].append(r)
for q, rs in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
    gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(rs, rs[1:])]
    big = [g for g in gaps if g > 20000]
    print('queue %s: %.1f launches/frame, busy %.3f ms/frame, first %.2f ms, last %.2f ms after the start; gaps > 20 us: %d, %.3f ms/frame; gaps <= 20 us: %.3f ms/frame'
          % (q, len(rs) / frames, busy / 1e6 / frames, (int(rs[0]['Start_Timestamp']) - t0) / 1e6, (int(rs[-1]['End_Timestamp']) - t0) / 1e6,
             len(big), sum(big) / 1e6 / frames, sum(g for g in gaps if 0 < g <= 20000) / 1e6 / frames))
    c, d = Counter(), defaultdict(float)
    for r in rs:
        c[r['Kernel_Name']] += 1
        d[r['Kernel_Name']] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    for name, us in sorted(d.items(), key=lambda kv: -kv[1])[:14]:
        print('   %6.1f /frame %8.1f us avg  %7.3f ms/frame  %s' % (c[name] / frames, us / c[name], us / 1e3 / frames, name[:100]))
    # what follows a big gap on this queue
    after = Counter(b['Kernel_Name'][:60] for a, b in zip(rs, rs[1:]) if int(b['Start_Timestamp']) - int(a['End_Timestamp']) > 20000)
    print('   after a gap > 20 us:', after.most_common(6))
