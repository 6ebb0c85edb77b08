"""Turn gpurun_out/prof_* (tools/profile_bench.sh) into the committed summaries under
profiles/: kernel stats, per-kernel PMC averages and traffic_voice_kernel.json (read by
bench.py for roofline.traffic).  Usage: python tools/summarize_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'profiles')
SRC = os.path.join(ROOT, 'gpurun_out')


def find(name, suffix):
    hits = glob.glob(os.path.join(SRC, 'prof_' + name, '**', '*' + suffix), recursive=True)
    if not hits:
        raise SystemExit('missing %s in prof_%s' % (suffix, name))
    return hits[0]


def pmc_table(name):
    """{kernel: {counter: (mean per launch, launches)}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(find(name, 'counter_collection.csv')) as fh:
        for row in csv.DictReader(fh):
            acc[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in cs.items()} for k, cs in acc.items()}


shutil.copy(find('stats', 'kernel_stats.csv'), os.path.join(OUT, '%s_kernel_stats_bench.csv' % tag))
shutil.copy(os.path.join(SRC, 'prof_stats.bench.json'), os.path.join(OUT, '%s_bench_under_rocprof.json' % tag))

tables = {n: pmc_table(n) for n in ('fetch', 'write', 'sq', 'sq2')}
with open(os.path.join(OUT, '%s_pmc_summary.csv' % tag), 'w') as fh:
    w = csv.writer(fh)
    w.writerow(['pass', 'kernel', 'counter', 'mean_per_launch', 'launches'])
    for n, t in tables.items():
        for k, cs in sorted(t.items()):
            for c, (mean, cnt) in sorted(cs.items()):
                w.writerow([n, k, c, '%.6g' % mean, cnt])


def voice(table):
    ks = [k for k in table if 'voice_kernel' in k]
    assert len(ks) == 1, ks
    return ks[0], table[ks[0]]


name, fetch = voice(tables['fetch'])
_, write = voice(tables['write'])
_, sq = voice(tables['sq'])
_, sq2 = voice(tables['sq2'])
bench = json.load(open(os.path.join(SRC, 'prof_stats.bench.json')))
cfg = bench['config']
units = cfg['voices_per_gpu'] * cfg['blocks_per_step'] * 256 / bench['roofline']['launches_per_step']
fetch_kb, write_kb = fetch['FETCH_SIZE'][0], write['WRITE_SIZE'][0]
summary = {
    'kernel': name,
    'round': tag,
    'voice_samples_per_launch': units,
    'FETCH_SIZE_KB_avg': fetch_kb,
    'WRITE_SIZE_KB_avg': write_kb,
    'correction': 'gfx950 (MI355X_MICROARCH.md, HBM/rocprofv3 section): FETCH_SIZE x2 (upper bound for '
                  '8 B/lane loads), WRITE_SIZE as read; both in KiB',
    'hbm_bytes_per_launch': (2. * fetch_kb + write_kb) * 1024.,
    'valu_instructions_per_voice_sample': sq['SQ_INSTS_VALU'][0] * 64. / units / 64.,
    'salu_instructions_per_voice_sample': sq['SQ_INSTS_SALU'][0] / (units / 64.),
    'lds_instructions_per_voice_sample': sq['SQ_INSTS_LDS'][0] / (units / 64.),
    'f64_add_mul_fma_per_voice_sample': [sq2[c][0] / (units / 64.) for c in
                                         ('SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64',
                                          'SQ_INSTS_VALU_FMA_F64')],
    'valu_busy_fraction_of_wave_cycles': sq['SQ_ACTIVE_INST_VALU'][0] / sq['SQ_WAVE_CYCLES'][0],
    'source': 'tools/profile_bench.sh: rocprofv3 --pmc passes in separate runs of bench.py --steps 2 --warmup 1',
}
# instruction counters are per wave: x/(units/64) = per wave-sample = per voice-sample per lane
summary['valu_instructions_per_voice_sample'] = sq['SQ_INSTS_VALU'][0] / (units / 64.)
with open(os.path.join(OUT, 'traffic_voice_kernel.json'), 'w') as fh:
    json.dump(summary, fh, indent=1)
print(json.dumps(summary, indent=1))
print(open(os.path.join(OUT, '%s_kernel_stats_bench.csv' % tag)).read())
