#!/bin/bash
# tools/collect_round.sh <tag>  -- the evidence of a round in one go, ON the GPU box from the repo root:
#   the GPU test tier, the bench line (incl. the sustained figure and the live PMC traffic), the same command under rocprofv3 --kernel-trace
#   --stats, PMC passes of the two headline kernels (128-bit x 1,024: SQ + TCC + GRBM; Uint5 x 512: SQ), all BASELINE configs, extended-table
#   timing, concurrent submitters (combine_bench at 64 / 256 threads).  (Replaces the per-round r04_*.sh / r05_final.sh scripts.)
# Everything lands in gpurun_out/<tag>/; copy what is to be kept into profiles/.
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --no-cpu-baseline --no-configs --sustained-steps 0 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err )
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_u5 -- python $R/tools/pmc_workload.py uint5 512 12 > $OUT/u5_under_rocprof.log 2>&1 )
find $OUT/stats_u5 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_uint5.csv \;
tools/prof_pmc.sh $TAG/pmc128 1024 > $OUT/pmc128.log 2>&1
tools/prof_pmc.sh $TAG/pmcu5 512 uint5 > $OUT/pmcu5.log 2>&1
python tools/measure_configs.py > $OUT/configs.json 2> $OUT/configs.err
python tools/ext_bench.py --batch 64 > $OUT/ext.log 2>&1
for T in 64 256; do for r in 1 2 3; do timeout 120 tools/combine_bench.bin $T quick; done; done > $OUT/combine.txt 2>&1
rm -rf $OUT/stats $OUT/stats_u5
ls -la $OUT
