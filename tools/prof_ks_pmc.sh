#!/bin/bash
# L2 counters of the matrix-core key switch (run ON the GPU box, from the repo root): tools/prof_ks_pmc.sh <tag> [batch]
set -u
TAG=${1:-kspmc}; B=${2:-1024}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python $R/tools/ks_workload.py $B > $OUT/$name.log 2>&1
  echo "pass $name rc=$?"; }
run tcc1 FETCH_SIZE
run tcc2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES
