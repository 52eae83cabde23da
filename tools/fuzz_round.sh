#!/bin/bash
# tools/fuzz_round.sh <tag> <minutes>: the differential fuzzer on the round's final library (incl. the seam entry points), its positive control
# (tools/build_variant_main.sh fuzzcontrol -DTFHE_FUZZ_CONTROL first; a control build is refused without TFHE_ALLOW_CONTROL_BUILD), and a
# soak of the concurrency tests on the lock-free combiner.
cd $GRAFT_REPO_ROOT; TAG=${1:-r06_g}; MIN=${2:-25}; O=gpurun_out/$TAG; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
V=$PWD/go-tfhe_amd/lib/variants/fuzzcontrol.so
{ echo "== a control build without the opt-in must be refused by the loader"; TFHE_HIP_LIB=$V python -c "
import __graft_entry__ as g
p = g.load_package()
try:
    p.load_library(); print('LOADED: the control build was NOT refused')
except Exception as e:
    print('refused:', str(e)[:160])"; } > $O/control.txt 2>&1
for S in 31 32 33; do
    echo "== control, seed $S" >> $O/control.txt
    TFHE_ALLOW_CONTROL_BUILD=1 TFHE_HIP_LIB=$V timeout 900 python tests/fuzz_gpu.py --seed $S --minutes 14 --log $O/control_seed$S.log; echo "rc=$? (1 = mismatch caught)" >> $O/control.txt
    tail -n 2 $O/control_seed$S.log >> $O/control.txt
done
timeout $((MIN * 60 + 300)) python tests/fuzz_gpu.py --seed 21 --minutes $MIN --log $O/fuzz_seed21.log; echo "fuzz rc=$?" > $O/fuzz.txt; tail -n 1 $O/fuzz_seed21.log >> $O/fuzz.txt
grep -c "kind=seams" $O/fuzz_seed21.log >> $O/fuzz.txt
for i in $(seq 1 20); do timeout 600 python -m pytest tests/test_gpu_concurrent.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 1; done > $O/soak.txt 2>&1
cat $O/control.txt $O/fuzz.txt; sort $O/soak.txt | uniq -c
