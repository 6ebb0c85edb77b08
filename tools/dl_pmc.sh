# Usage (GPU box): bash tools/dl_pmc.sh -- HBM read / write counters of Filter -> Delay -> Gain, per delay-time mode
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out && cd /tmp && export TMPDIR=/tmp
for mode in random equal; do
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_dl_${mode}_$c
  KLANG_BENCH_DELAY=$mode timeout -s KILL 150 rocprofv3 --pmc $c -d $R/gpurun_out/pmc_dl_${mode}_$c -o out --output-format csv -- python3 $R/tools/bench_configs.py 3delay > $R/gpurun_out/pmc_dl_${mode}_$c.log 2>&1 || { tail -5 $R/gpurun_out/pmc_dl_${mode}_$c.log; exit 1; }
done
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for mode in ('random', 'equal'):
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        f = glob.glob('gpurun_out/pmc_dl_%s_%s/**/*counter_collection.csv' % (mode, c), recursive=True)[0]
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if 'voice_fx_kernel' in row['Kernel_Name'] and row['Counter_Name'] == c:
                acc[c].append(float(row['Counter_Value']))
        v = acc[c]
        print(mode, c, 'launches', len(v), 'mean KiB %.0f -> GB %.2f' % (sum(v) / len(v), sum(v) / len(v) * 1024 / 1e9))
PY
