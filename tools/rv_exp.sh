set -e
mkdir -p gpurun_out
: > gpurun_out/rv_exp.jsonl
timeout -k 10 300 python -u -m pytest tests/test_gpu_reverb_scale.py tests/test_zz_no_alloc.py -x -q --timeout 180 --timeout-method thread > gpurun_out/rv_tests.log 2>&1 || { tail -30 gpurun_out/rv_tests.log; exit 1; }
tail -2 gpurun_out/rv_tests.log
for occ in 3 2; do
for vpw in 1 2 4; do
KLMI_REVERB_OCC=$occ KLMI_REVERB_VPW=$vpw timeout -k 10 120 python tools/bench_configs.py 4f32 4bigf32 >> gpurun_out/rv_exp.jsonl
done
done
python - <<'PY'
import json
for l in open('gpurun_out/rv_exp.jsonl'):
    d=json.loads(l)
    print(d['config'][:14], d['knobs'], 'kernel %.3f ms call %.3f ms frac %.3f whole %.3f' % (d['reverb_kernel_ms'], d['wall_ms'], d['roofline']['frac'], d['roofline']['whole_call_frac']))
PY
