#!/bin/bash
# build_variants.sh name:"-Dflags" ...  -> tools/ubench/bin/vv_<name>
cd "$(dirname "$0")/../.." && mkdir -p tools/ubench/bin
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I klang_amd/csrc -Wno-unused-result -Wno-unused-value $flags tools/ubench/voice_variants.hip -o tools/ubench/bin/vv_$name &
done
wait
ls -la tools/ubench/bin
