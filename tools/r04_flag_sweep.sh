#!/bin/bash
# tools/r04_flag_sweep.sh <tag> <libs...>: compiler-option variants of the blind-rotate units on four workloads (one box, interleaved)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for pb in "128 1024" "128 1" "uint5 512" "uint2 2048"; do
  set -- $pb "${@:1}"
  P=$1; B=$2; shift 2
  echo "== params $P batch $B" >> $OUT/sweep.txt
  python tools/ab_bench.py --params $P --batch $B --rounds 3 --launches 8 "$@" >> $OUT/sweep.txt 2>&1
done
cat $OUT/sweep.txt
