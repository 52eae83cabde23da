#!/bin/bash
# tools/build_variant_main.sh <name> [extra flags for tfhe_hip.hip ...] -> go-tfhe_amd/lib/variants/<name>.so
# (build_variant.sh passes its flags to blind_rotate.hip; this one to the other translation unit)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p go-tfhe_amd/lib/variants /tmp/var_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
/opt/rocm/bin/hipcc $F "$@" -c go-tfhe_amd/csrc/tfhe_hip.hip -o /tmp/var_$name/a.o 2>/dev/null &
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-sched-strategy=max-ilp -c go-tfhe_amd/csrc/blind_rotate.hip -o /tmp/var_$name/b.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_$name/a.o /tmp/var_$name/b.o -o go-tfhe_amd/lib/variants/$name.so
echo built go-tfhe_amd/lib/variants/$name.so
