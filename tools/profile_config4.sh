#!/bin/bash
# Usage (GPU box, via gpurun): bash tools/profile_config4.sh [4|4f32|4big|4bigf32]
# rocprofv3 over BASELINE configs[3] (tools/bench_configs.py 4): kernel trace + stats, then
# HBM read / write counters, each pass in a run of its own.  Results: gpurun_out/prof4_*.
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-4}
mkdir -p $R/gpurun_out && cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprofv3 options...
  local name=$1; shift
  rm -rf $R/gpurun_out/prof4_${W}_$name
  timeout -s KILL 150 rocprofv3 "$@" -d $R/gpurun_out/prof4_${W}_$name -o out --output-format csv -- python3 $R/tools/bench_configs.py $W > $R/gpurun_out/prof4_${W}_$name.log 2>&1 || { tail -5 $R/gpurun_out/prof4_${W}_$name.log; exit 1; }
}
run stats --kernel-trace --stats &&
run fetch --pmc FETCH_SIZE &&
run write --pmc WRITE_SIZE &&
echo profiled $W
