#!/bin/bash
# tools/r04_fold8_ab.sh <tag> <libs...>: FMA-folded radix-8 forward levels -- bit-identity at the exact sets, interleaved A/B at the headline
# shapes (128-bit x 1,024 / x 768 / x 512, Uint5 x 512) with shader clock and power sampled during the launches
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
python tools/ab_equal.py --params 128 --batches 300,700,1024 "$@" > $OUT/equal.txt 2>&1
python tools/ab_equal.py --params uint5 --batches 64,512 "$@" >> $OUT/equal.txt 2>&1
for B in 1024 768 512; do
  echo "== 128-bit batch $B" >> $OUT/ab.txt
  python tools/ab_bench.py --batch $B --rounds 4 --launches 30 "$@" >> $OUT/ab.txt 2>&1
done
echo "== uint5 batch 512" >> $OUT/ab.txt
python tools/ab_bench.py --params uint5 --batch 512 --rounds 4 --launches 30 "$@" >> $OUT/ab.txt 2>&1
cat $OUT/equal.txt $OUT/ab.txt
