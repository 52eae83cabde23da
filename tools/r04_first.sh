set -x
mkdir -p gpurun_out/r04a
python go-tfhe_amd/telemetry.py > gpurun_out/r04a/telemetry.txt 2>&1
timeout 600 python bench.py > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err
echo "bench rc=$?"
TFHE_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --config5-gates 32768 > gpurun_out/r04a/weak_n2share.json 2> gpurun_out/r04a/weak_n2share.err
echo "n2share rc=$?"
timeout 300 python bench.py --mode sharded --workload mixed --gates 65536 --steps 1 > gpurun_out/r04a/sh_mixed_rccl1.json 2> gpurun_out/r04a/sh_mixed_rccl1.err
echo "rccl1 rc=$?"
tools/prof_pmc.sh r04a/pmcu5 512 uint5 > gpurun_out/r04a/pmcu5.log 2>&1
tail -5 gpurun_out/r04a/bench.err gpurun_out/r04a/weak_n2share.err gpurun_out/r04a/sh_mixed_rccl1.err
cat gpurun_out/r04a/telemetry.txt
