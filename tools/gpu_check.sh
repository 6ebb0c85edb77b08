#!/bin/bash
# Usage (on the GPU box, via gpurun): bash tools/gpu_check.sh [bench-steps]
# GPU parity tests, then smoke, then a short bench; each step under its own timeout.
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -4 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 400 python bench.py --steps ${1:-3} --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench.json'))
r = d['roofline']
print('value %.4g %s  ms/step %.2f  kernel ms %.3f x%d  hbm frac %.4f valu frac %.3f' % (
    d['value'], d['unit'], d['ms_per_step'], r['avg_launch_ms'], r['launches_per_step'], r['frac'], r['valu_frac']))
if 'cpu_baseline' in d:
    print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
