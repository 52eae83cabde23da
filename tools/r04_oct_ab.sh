#!/bin/bash
# tools/r04_oct_ab.sh <tag> <libs...>: bit-identity (SHA-256 of accumulators and bootstrapped samples) and interleaved A/B timing of
# library variants at the small-launch sizes (eight-wave kernel), on ONE box
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
python tools/ab_equal.py --params 128 --batches 1,7,128,256,300 "$@" > $OUT/equal.txt 2>&1
for B in 1 128 256; do
  echo "== batch $B" >> $OUT/ab.txt
  python tools/ab_bench.py --batch $B --rounds 3 --launches 10 "$@" >> $OUT/ab.txt 2>&1
done
cat $OUT/equal.txt $OUT/ab.txt
