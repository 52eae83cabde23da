cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_b; O=gpurun_out/r06_b
nproc > $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>&1; grep -m1 "model name" /proc/cpuinfo >> $O/host.txt
g++ -O2 -std=c++17 tools/ubench_wake.cpp -o /tmp/ubench_wake -lpthread
for T in 63 255; do for m in 0 1; do /tmp/ubench_wake $T $m; done; done > $O/wake.txt 2>&1
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
for T in 1 16 64 128 256; do for r in 1 2 3; do tools/combine_bench.bin $T; done; done > $O/combine_base.txt 2>&1
cat $O/host.txt $O/wake.txt $O/combine_base.txt
