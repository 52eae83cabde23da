cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_h; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
for Q in ${QS:-150}; do for T in ${TS:-2 16 64 256}; do for r in 1 2 3; do echo -n "quiet=$Q "; timeout 120 tools/combine_bench.bin $T quick $Q; done; done; done > $O/quiet_sweep.txt 2>&1
cat $O/quiet_sweep.txt
