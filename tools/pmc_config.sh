#!/bin/bash
# Usage (GPU box): bash tools/pmc_config.sh <bench_configs name>  -- SQ counters of one secondary config
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out && cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_$1
timeout -s KILL 150 rocprofv3 --pmc ${PMC:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU} -d $R/gpurun_out/pmc_$1 -o out --output-format csv -- python3 $R/tools/bench_configs.py $1 > $R/gpurun_out/pmc_$1.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$1.log; exit 1; }
echo ok
