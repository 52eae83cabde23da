cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_d; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
for T in 2 16 64 128 256; do for r in 1 2 3; do timeout 120 tools/combine_bench.bin $T quick; done; done > $O/combine.txt 2>&1
cat $O/combine.txt
