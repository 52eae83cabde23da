#!/bin/bash
# tools/r05_final.sh <tag>: round-5 evidence in one go (GPU box, repo root): the GPU tests (incl. the vectors computed by the reference's
# own source, tests/test_goref_vectors.py), the bench line, the same command under rocprofv3 --stats, the PMC passes of the two headline
# kernels, a sustained run, the multi-rank dry runs, the in-process multi-GPU form (clone) through the C++ mirror.
TAG=${1:-r05z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
python -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; grep -E "passed|failed" $OUT/pytest.txt
python -m pytest tests/test_goref_vectors.py -m gpu -q -rA 2>&1 | grep -E "PASSED|FAILED|SKIPPED|passed|failed" > $OUT/pytest_goref.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --no-cpu-baseline --no-configs > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err )
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_u5 -- python $R/tools/pmc_workload.py uint5 512 12 > $OUT/u5_under_rocprof.log 2>&1 )
find $OUT/stats_u5 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_uint5.csv \;
tools/prof_pmc.sh $TAG/pmc128 1024 > $OUT/pmc128.log 2>&1
tools/prof_pmc.sh $TAG/pmcu5 512 uint5 > $OUT/pmcu5.log 2>&1
python bench.py --steps 3000 --warmup 50 --no-configs --no-cpu-baseline > $OUT/sustained.json 2> $OUT/sustained.err; echo "sustained rc=$?"
TFHE_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --config5-gates 65536 > $OUT/weak_n2share.json 2> $OUT/weak_n2share.err; echo "n2share rc=$?"
timeout 300 python bench.py --mode sharded --workload mixed --gates 131072 --steps 1 > $OUT/sh_mixed_rccl1.json 2> $OUT/sh_mixed_rccl1.err; echo "rccl1 rc=$?"
timeout 300 python bench.py --mode sharded --workload adder --steps 5 > $OUT/sh_adder_rccl1.json 2> $OUT/sh_adder_rccl1.err; echo "rccl1 adder rc=$?"
tests/cpp/test_host_mirror.bin > $OUT/cpp_host_mirror.txt 2>&1; tail -2 $OUT/cpp_host_mirror.txt
tools/combine_bench.bin 256 > $OUT/combine.txt 2>&1
python tools/measure_configs.py > $OUT/configs.json 2> $OUT/configs.err
rm -rf $OUT/stats $OUT/stats_u5 $OUT/pmc128/*/ $OUT/pmcu5/*/
ls $OUT
