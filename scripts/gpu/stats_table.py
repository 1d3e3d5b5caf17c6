"""Compact per-step table of a rocprofv3 kernel_stats.csv: python stats_table.py FILE STEPS [N]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', name)
    if m:
        n = int(m.group(1))
        rest = name[len(m.group(0)):]
        name = rest[:n] + ' ' + rest[n:n + 40]
    if 'rocprim' in name:
        k = re.findall(r'(radix_sort_\w+|merge_sort_\w+|onesweep\w*|lookback_scan\w*|\w*histogram\w*|partition\w*|transform\w*|scan\w*)', name)
        name = 'rocprim:' + (k[0] if k else name[:40]) + (':' + re.findall(r'lambda\(auto:1\)#(\d)', name)[-1] if 'lambda(auto:1)#' in name else '')
    return name[:78]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = float(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    agg = {}
    for r in rows:
        k = short(r['Name'])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += int(r['Calls'])
        a[1] += float(r['TotalDurationNs'])
    tot = sum(a[1] for a in agg.values())
    print('total %.3f ms/step, %d launches/step' % (tot / steps / 1e6, sum(a[0] for a in agg.values()) / steps))
    for k, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%8.3f ms %6.1f calls %7.1f us  %s' % (ns / steps / 1e6, c / steps, ns / c / 1e3, k))


if __name__ == '__main__':
    main()
