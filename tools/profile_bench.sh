#!/bin/bash
# Usage (GPU box, via gpurun): bash tools/profile_bench.sh
# rocprofv3 passes over the default bench.py workload, each in a run of its own:
#   kernel trace + stats, then PMC passes (HBM read, HBM write, SQ instruction counters).
# Results land in gpurun_out/prof_*; tools/summarize_profiles.py turns them into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out && cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extra"
run() {  # name, rocprofv3 options...
  local name=$1; shift
  rm -rf $R/gpurun_out/prof_$name
  timeout -s KILL 150 rocprofv3 "$@" -d $R/gpurun_out/prof_$name -o out --output-format csv -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$name.log 2>&1 || { tail -5 $R/gpurun_out/prof_$name.log; exit 1; }
  grep '^{' $R/gpurun_out/prof_$name.log > $R/gpurun_out/prof_$name.bench.json
}
run stats --kernel-trace --stats &&
run fetch --pmc FETCH_SIZE &&
run write --pmc WRITE_SIZE &&
run sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU &&
run sq2 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS &&
echo profiled
