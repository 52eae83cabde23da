#!/bin/bash
# tools/r04_sched_matrix.sh <tag> <libs...>: machine-scheduler options of the blind-rotate translation unit, per kernel family (one box)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { echo "== params $1 batch $2" >> $OUT/matrix.txt; python tools/ab_bench.py --params $1 --batch $2 --rounds 3 --launches 8 "$@" 2>&1 | grep -v "^==" >> $OUT/matrix.txt; }
for B in 1 256 512 768 1024; do echo "== params 128 batch $B" >> $OUT/matrix.txt; python tools/ab_bench.py --params 128 --batch $B --rounds 3 --launches 8 "$@" >> $OUT/matrix.txt 2>&1; done
for B in 1 512; do echo "== params uint5 batch $B" >> $OUT/matrix.txt; python tools/ab_bench.py --params uint5 --batch $B --rounds 3 --launches 8 "$@" >> $OUT/matrix.txt 2>&1; done
echo "== params uint2 batch 2048" >> $OUT/matrix.txt; python tools/ab_bench.py --params uint2 --batch 2048 --rounds 3 --launches 8 "$@" >> $OUT/matrix.txt 2>&1
for B in 256 1024; do echo "== params uint3 batch $B" >> $OUT/matrix.txt; python tools/ab_bench.py --params uint3 --batch $B --rounds 3 --launches 8 "$@" >> $OUT/matrix.txt 2>&1; done
echo "== params uint1 batch 1024" >> $OUT/matrix.txt; python tools/ab_bench.py --params uint1 --batch 1024 --rounds 3 --launches 8 "$@" >> $OUT/matrix.txt 2>&1
cat $OUT/matrix.txt
