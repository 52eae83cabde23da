#!/bin/bash
# tools/build_variant.sh <name> [extra flags for the blind-rotate translation units ...] -> go-tfhe_amd/lib/variants/<name>.so
# (the units and their machine-scheduler options are go-tfhe_amd/build.py's; N2048FLAGS in the environment adds flags to the N = 2048 unit only)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p go-tfhe_amd/lib/variants /tmp/var_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
ILP="-mllvm -amdgpu-sched-strategy=max-ilp"
/opt/rocm/bin/hipcc $F -c go-tfhe_amd/csrc/tfhe_hip.hip -o /tmp/var_$name/a.o 2>/dev/null &
/opt/rocm/bin/hipcc $F $ILP "$@" -c go-tfhe_amd/csrc/blind_rotate.hip -o /tmp/var_$name/b.o 2>/dev/null &
/opt/rocm/bin/hipcc $F $ILP -mllvm -enable-post-misched=0 "$@" -c go-tfhe_amd/csrc/blind_rotate_oct.hip -o /tmp/var_$name/c.o 2>/dev/null &
/opt/rocm/bin/hipcc $F $N2048FLAGS "$@" -c go-tfhe_amd/csrc/blind_rotate_n2048.hip -o /tmp/var_$name/d.o 2>/dev/null &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_$name/{a,b,c,d}.o -o go-tfhe_amd/lib/variants/$name.so
echo built go-tfhe_amd/lib/variants/$name.so
