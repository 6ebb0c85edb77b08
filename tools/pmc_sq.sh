#!/bin/bash
# Usage (GPU box): bash tools/pmc_sq.sh  -- SQ issue/wait counters of the voice kernel, split off/on
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for split in 0 1; do
  export KLMI_SPLIT=$split
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmc_sq_$split -o sq --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_sq_$split.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmc_sq2_$split -o sq --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_sq2_$split.log 2>&1 || exit 1
done
echo done
