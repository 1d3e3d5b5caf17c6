"""profiles/rNN_pmc_conv_apply.json from a final-run directory: python scripts/gpu/pmc_record.py gpurun_out/final5_a r05 'script tag'"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                    # source_digest: the record is void once the kernel sources change
from lidal_amd import backend as B
d, rnd, how = sys.argv[1], sys.argv[2], sys.argv[3]
txt = open(d + '/pmc_summary.txt').read()

def section(name):
    """the block of pmc_summary.txt that belongs to the pass directory `name` (the summary also holds the step passes
    sfetch/ and swrite/, whose first WRITE_SIZE line is another kernel's: the first three round-6 records quoted it)"""
    m = re.search(r'^== %s/.*?$(.*?)(?=^== |\Z)' % re.escape(name), txt, re.S | re.M)
    return m.group(1)


KERNEL = 'conv_lean_kernelIDF16bLi6ELi192'
f_line = [l for l in section('fetch').splitlines() if KERNEL in l and 'FETCH_SIZE' in l][0]
w_line = [l for l in section('write').splitlines() if KERNEL in l and 'WRITE_SIZE' in l][0]
fetch = float(re.search(r'mean=([\d.e+]+)', f_line).group(1))
n = int(re.search(r'n=(\d+)', f_line).group(1))
write = float(re.search(r'mean=([\d.e+]+)', w_line).group(1))
assert int(re.search(r'n=(\d+)', w_line).group(1)) == n
row = [r for r in csv.DictReader(open(d + '/roof_kernel_stats.csv')) if 'conv_lean_kernelIDF16bLi6ELi192' in r['Name']][0]
line = json.load(open(d + '/bench_line.json'))['roofline']
traffic = int(round((2 * fetch + write) * 1024))
algo = line['algorithmic_bytes_per_launch']
rec = {'workload': {'rows': line['rows'], 'rules': line['rules'], 'dtype': 'bf16', 'layer': 'k3 s1 96->96',
                    'kernel': 'conv_lean_kernel<bf16,6,192,8>'},
       'library': {'version': int(B.lib_handle().lidal_version()), 'sources_sha16': bench.source_digest(('conv_img.hip',)),
                   'sources': ['conv_img.hip']},
       'command': 'rocprofv3 --pmc FETCH_SIZE (and a second pass --pmc WRITE_SIZE) -- python3 bench.py --roofline-only  (%s)' % how,
       'FETCH_SIZE_KB_mean_of_%d' % n: fetch, 'WRITE_SIZE_KB_mean_of_%d' % n: write,
       'correction': 'MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane reads; WRITE_SIZE is exact',
       'traffic_bytes': traffic, 'algorithmic_bytes': algo, 'ratio': round(traffic / algo, 3),
       'kernel_stats': 'profiles/%s_roofline_only_kernel_stats.csv: %s launches, %.1f us average (bench.py\'s own HIP events in the default run of the same box: %.1f)'
                       % (rnd, row['Calls'], float(row['AverageNs']) / 1e3, line['launch_us'])}
json.dump(rec, open('profiles/%s_pmc_conv_apply.json' % rnd.split('_')[0], 'w'), indent=1)
print(rec)
