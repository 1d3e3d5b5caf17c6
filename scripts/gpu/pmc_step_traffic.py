"""HBM-side traffic of one training step from the L2's fabric counters: sums FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, one
counter per pass; KB as the tool reports them) over every dispatch of `bench.py --steps S`, cut at the optimizer launches
like scripts/gpu/steady_counts.py, per step and by kernel.  The gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE
counts 64-byte requests as 32: x 2) is applied to the fetch side.
usage: pmc_step_traffic.py FETCH_CSV WRITE_CSV [ROWS DTYPE OUT.json]   (with the last three: the record bench.py reads)"""
import csv
import sys
from collections import defaultdict


def per_step(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r.get('Counter_Name') == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    adam = [i for i, r in enumerate(rows) if 'FusedAdam' in r['Kernel_Name'] or 'FusedOptimizer' in r['Kernel_Name']]
    bursts = []
    for i in adam:
        if not bursts or i - bursts[-1][1] > 20:
            bursts.append([i, i])
        bursts[-1][1] = i
    skip = 3                        # (host_calls() of bench.py runs a second model at the end)
    n_steps = min(len(bursts) - 1 - skip, 5)
    lo, hi = bursts[len(bursts) - 1 - skip - n_steps][0], bursts[len(bursts) - 1 - skip][0]
    by = defaultdict(float)
    for r in rows[lo:hi]:
        by[r['Kernel_Name'][:70]] += float(r['Counter_Value'])
    return {k: v / n_steps for k, v in by.items()}, n_steps


fetch, ns = per_step(sys.argv[1], 'FETCH_SIZE')
write, _ = per_step(sys.argv[2], 'WRITE_SIZE')
tf, tw = 2 * sum(fetch.values()) * 1024 / 1e9, sum(write.values()) * 1024 / 1e9
print('per step (%d steps): fetched %.2f GB (2 x FETCH_SIZE), written %.2f GB, together %.2f GB' % (ns, tf, tw, tf + tw))
names = sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0)))
for k in names[:30]:
    print('   %8.1f MB fetched %8.1f MB written   %s' % (2 * fetch.get(k, 0) * 1024 / 1e6, write.get(k, 0) * 1024 / 1e6, k))

if len(sys.argv) > 5:
    import json, os
    ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, ROOT)
    import bench
    from lidal_amd import backend as B
    rec = {'workload': {'rows': int(sys.argv[3]), 'dtype': sys.argv[4], 'what': 'the headline step of bench.py (5 scans, tables on the second stream)'},
           'library': {'version': int(B.lib_handle().lidal_version()), 'sources_sha16': bench.source_digest(bench.KERNEL_SOURCES),
                       'sources': list(bench.KERNEL_SOURCES)},
           'command': 'rocprofv3 --pmc FETCH_SIZE (and a second pass --pmc WRITE_SIZE) -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline --no-families --no-variants',
           'correction': 'MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane reads; WRITE_SIZE is exact',
           'steps_averaged': ns, 'fetched_GB': round(tf, 3), 'written_GB': round(tw, 3), 'traffic_GB': round(tf + tw, 3),
           'by_kernel_MB': {k: round((2 * fetch.get(k, 0) + write.get(k, 0)) * 1024 / 1e6, 1) for k in names[:40]}}
    json.dump(rec, open(sys.argv[5], 'w'), indent=1)
