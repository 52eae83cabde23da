# round 5: HBM traffic of the wide key switch (Uint5 x 512) with and without the per-XCD partial sums: FETCH_SIZE and WRITE_SIZE in
# separate passes (one counter per pass, kernel trace only), per kernel of the key-switch path
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r05d
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for MODE in 1 0; do
 for CTR in FETCH_SIZE WRITE_SIZE; do
  KS_XCD_SUM=$MODE rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $R/gpurun_out/r05d/m${MODE}_$CTR -- python $R/tools/pmc_workload.py uint5 512 > $R/gpurun_out/r05d/m${MODE}_$CTR.log 2>&1
 done
done
python3 - <<P
import csv, glob, collections
out = []
for mode in (1, 0):
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob("$R/gpurun_out/r05d/m%d_%s/**/*counter_collection.csv" % (mode, ctr), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == ctr:
                    name = row["Kernel_Name"].split("(")[0]
                    if any(k in name for k in ("k_keyswitch_wide", "k_ks_xsum_reduce", "k_ks_init", "fillBuffer", "Memset", "memset")):
                        acc[name].append(float(row["Counter_Value"]))
        for name, v in sorted(acc.items()):
            mult = 2 if ctr == "FETCH_SIZE" else 1      # gfx950: FETCH_SIZE counts 64-byte halves of the 128-byte requests (MI355X guide) -> x 2
            out.append("ks_xcd_sum=%d %-11s %-60s %8.1f KiB raw per launch x %d launches -> %7.1f MB" % (mode, ctr, name[:60], sum(v) / len(v), len(v), mult * 1024 * sum(v) / len(v) / 1e6))
print("\n".join(out))
open("$R/gpurun_out/r05d/summary.txt", "w").write("\n".join(out) + "\n")
P
