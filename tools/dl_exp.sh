set -e
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_effects.py tests/test_gpu_kernel_agreement.py tests/test_gpu_graph.py tests/test_gpu_examples.py tests/test_zz_no_alloc.py -x -q --timeout 180 --timeout-method thread > gpurun_out/dl_tests.log 2>&1 || { tail -40 gpurun_out/dl_tests.log; exit 1; }
tail -2 gpurun_out/dl_tests.log
: > gpurun_out/dl_exp.jsonl
for m in random few equal; do
KLANG_BENCH_DELAY=$m timeout -k 10 120 python tools/bench_configs.py 3delay >> gpurun_out/dl_exp.jsonl
done
KLMI_FX_KERNEL=0 timeout -k 10 120 python tools/bench_configs.py 3delay >> gpurun_out/dl_exp.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/dl_exp.jsonl'):
    d=json.loads(l)
    print(d['config'][:110], '%.3g' % d['value'], 'kernel %.2f ms' % d['kernel_ms'])
PY
