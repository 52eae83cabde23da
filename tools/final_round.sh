# tools/final_round.sh [tag]: the round-end collection (collect_round.sh + sanitizers + 2-rank shared-GPU dry run + smoke)
cd $GRAFT_REPO_ROOT; TAG=${1:-r06_t}
tools/collect_round.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
O=gpurun_out/$TAG
for san in thread control address; do timeout 600 tools/asan_host_check.sh run $san > $O/san_$san.txt 2>&1; echo "rc=$?" >> $O/san_$san.txt; done
# the N > 1 path of bench.py as a 2-rank dry run sharing the one GPU (gloo through host memory: RCCL refuses two ranks on one device)
TFHE_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 5 --warmup 2 --sustained-steps 100 --config5-gates 65536 > $O/weak_n2share.json 2> $O/weak_n2share.err; echo "n2share rc=$?" >> $O/weak_n2share.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
tail -n 3 $O/pytest_gpu.txt; head -c 400 $O/bench.json; echo; tail -n 2 $O/san_*.txt; tail -n 2 $O/weak_n2share.err; head -c 300 $O/weak_n2share.json; echo; cat $O/smoke.txt | tail -n 1; cat $O/combine.txt | cut -c1-200
