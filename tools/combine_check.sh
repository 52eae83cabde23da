#!/bin/bash
# tools/combine_check.sh <tag>: the lock-free combiner and the seam entry points on the GPU box -- tests, the concurrent-submitter
# bench at 1 ... 256 threads (3 runs each), the ThreadSanitizer / AddressSanitizer builds (tools/asan_host_check.sh build ... first, here).
cd $GRAFT_REPO_ROOT; TAG=${1:-r06_c}; O=gpurun_out/$TAG; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_seams.py tests/test_gpu_concurrent.py tests/test_gpu_cpp_host.py tests/test_gpu_clone.py tests/test_gpu_path.py \
    tests/test_gpu_circuits.py tests/test_gpu_abi_misuse.py -m gpu -q -x -p no:cacheprovider > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
for T in 1 2 16 64 128 256; do for r in 1 2 3; do timeout 120 tools/combine_bench.bin $T; done; done > $O/combine.txt 2>&1
for T in 64 256; do timeout 120 tools/combine_bench.bin $T pbs; done >> $O/combine.txt 2>&1
for san in thread control address; do
    [ -d go-tfhe_amd/lib/variants/${san}san ] && { timeout 600 tools/asan_host_check.sh run $san > $O/san_$san.txt 2>&1; echo "rc=$?" >> $O/san_$san.txt; }
done
tail -n 15 $O/pytest.txt; cat $O/combine.txt; tail -n 4 $O/san_*.txt
