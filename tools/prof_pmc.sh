#!/bin/bash
# PMC passes for the two path kernels (run ON the GPU box, from the repo root):
#   tools/prof_pmc.sh <tag> [batch] [params]      (params given: tools/pmc_workload.py at that set)
# Each pass is its own rocprofv3 run with --kernel-trace only (no other trace domains).
set -u
TAG=${1:-pmc}; B=${2:-1024}; PSET=${3:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
WORK="python $R/tests/gpu_first_light.py $B"
[ -n "$PSET" ] && WORK="python $R/tools/pmc_workload.py $PSET $B"
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- \
      $WORK > $OUT/$name.log 2>&1
  echo "pass $name rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
