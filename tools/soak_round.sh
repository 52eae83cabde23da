#!/bin/bash
# tools/soak_round.sh <tag> <fuzz minutes> <suite repeats> [seed]: a long fuzzer run with a fresh seed and N repeats of the whole GPU test tier on the final library
cd $GRAFT_REPO_ROOT; TAG=${1:-soak}; MIN=${2:-40}; REP=${3:-5}; SEED=${4:-22}; O=gpurun_out/$TAG; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout $((MIN * 60 + 300)) python tests/fuzz_gpu.py --seed $SEED --minutes $MIN --log $O/fuzz_seed$SEED.log; echo "fuzz rc=$?" > $O/fuzz.txt; tail -n 1 $O/fuzz_seed$SEED.log >> $O/fuzz.txt
for i in $(seq 1 $REP); do timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -n 1; done > $O/suite_soak.txt 2>&1
cat $O/fuzz.txt $O/suite_soak.txt
