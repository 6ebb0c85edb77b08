set -e
mkdir -p gpurun_out
: > gpurun_out/rv_exp.jsonl
KLMI_REVERB_FUSED=0 timeout -k 10 120 python tools/bench_configs.py 4 4big >> gpurun_out/rv_exp.jsonl
for vpw in 1 2 4; do
KLMI_REVERB_VPW=$vpw timeout -k 10 120 python tools/bench_configs.py 4 4big >> gpurun_out/rv_exp.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/rv_exp.jsonl'):
    d=json.loads(l)
    print(d['config'][:14], d['knobs'], 'kernel %.3f ms call %.3f ms frac %.3f whole %.3f' % (d['reverb_kernel_ms'], d['wall_ms'], d['roofline']['frac'], d['roofline']['whole_call_frac']))
PY
