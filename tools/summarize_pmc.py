"""Per wave-sample instruction counts of the voice kernel in one `tools/pmc_config.sh` run.

    python tools/summarize_pmc.py gpurun_out/pmc_3fm [samples_per_launch]

Reads rocprofv3's counter_collection.csv, keeps the dispatches of the voice kernels
(voice_kernel / voice_mod_kernel / voice_pipe_kernel), and divides every SQ counter by
SQ_WAVES x samples (samples = n_blocks * 256 of tools/bench_configs.py: 128 * 256).
"""
import collections
import csv
import json
import sys


def main():
    d = sys.argv[1]
    samples = int(sys.argv[2]) if len(sys.argv) > 2 else 128 * 256
    rows = list(csv.DictReader(open(d + '/out_counter_collection.csv')))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    regs = {}
    for r in rows:
        name = r['Kernel_Name']
        if 'voice_' not in name:
            continue
        short = name.split('(')[0].replace('void ', '')
        per[short][r['Counter_Name']].append(float(r['Counter_Value']))
        regs[short] = dict(vgpr=int(r['VGPR_Count']), agpr=int(r['Accum_VGPR_Count']),
                           grid=int(r['Grid_Size']))
    for k, c in per.items():
        waves = regs[k]['grid'] / 64
        out = dict(kernel=k, launches=len(next(iter(c.values()))), waves=waves, **regs[k])
        for name, vals in sorted(c.items()):
            med = sorted(vals)[len(vals) // 2]
            if name == 'SQ_WAVES':
                out['SQ_WAVES'] = med
            else:
                out[name + '_per_wave_sample'] = round(med / waves / samples, 3)
        print(json.dumps(out))


if __name__ == '__main__':
    main()
