"""gpurun_out/prof4_<W>_* (tools/profile_config4.sh) -> profiles/<tag>_config4_<W>_*.  Usage:
python tools/summarize_config4.py r02 4f32"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, W = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, OUT = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')


def find(name, suffix):
    return glob.glob(os.path.join(SRC, 'prof4_%s_%s' % (W, name), '**', '*' + suffix), recursive=True)[0]


shutil.copy(find('stats', 'kernel_stats.csv'), os.path.join(OUT, '%s_config4_%s_kernel_stats.csv' % (tag, W)))
counters = {}
for name in ('fetch', 'write'):
    acc = collections.defaultdict(list)
    with open(find(name, 'counter_collection.csv')) as fh:
        for row in csv.DictReader(fh):
            if 'reverb_' in row['Kernel_Name'] and 'short' not in row['Kernel_Name']:
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
    for c, v in acc.items():
        counters[c] = (sum(v) / len(v), len(v))
bench = json.loads([l for l in open(os.path.join(SRC, 'prof4_%s_stats.log' % W)) if l.startswith('{')][0])
launches = bench['reverb_launches']
cfg = bench['config']
n_ch = int(cfg.split(':')[1].split()[0])
n_blocks = int(cfg.split('Mixer,')[1].split()[0])
per_sample = bench['roofline']['algorithmic_bytes_per_channel_sample']
algo = n_ch * n_blocks * 256 * per_sample / launches
fetch_kb, write_kb = counters['FETCH_SIZE'][0], counters['WRITE_SIZE'][0]
summary = {
    'round': tag, 'workload': cfg, 'kernel': bench['roofline']['kernel'], 'launches_per_call': launches,
    'algorithmic_bytes_per_launch': algo,
    'FETCH_SIZE_KB_avg': fetch_kb, 'WRITE_SIZE_KB_avg': write_kb,
    'correction': 'gfx950 (MI355X_MICROARCH.md, HBM section): FETCH_SIZE x2 for 16 B/lane streaming '
                  'reads, WRITE_SIZE as read; both in KiB',
    'hbm_bytes_per_launch': (2. * fetch_kb + write_kb) * 1024.,
    'traffic_over_algorithmic': (2. * fetch_kb + write_kb) * 1024. / algo,
    'bench_under_rocprof': bench,
    'source': 'tools/profile_config4.sh %s: rocprofv3 --kernel-trace --stats and --pmc FETCH_SIZE / '
              'WRITE_SIZE in separate runs of tools/bench_configs.py %s' % (W, W),
}
with open(os.path.join(OUT, '%s_config4_%s_traffic.json' % (tag, W)), 'w') as fh:
    json.dump(summary, fh, indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != 'bench_under_rocprof'}, indent=1))
print(open(os.path.join(OUT, '%s_config4_%s_kernel_stats.csv' % (tag, W))).read()[:1500])
