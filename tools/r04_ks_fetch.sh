mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r04h
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 256 512; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r04h/f$B -- python $R/tools/pmc_workload.py uint5 $B > $R/gpurun_out/r04h/f$B.log 2>&1
  python3 - <<P
import csv, glob
v = []
for f in glob.glob("$R/gpurun_out/r04h/f$B/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_keyswitch_wide" in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
            v.append(float(row["Counter_Value"]))
print("B=$B FETCH_SIZE KiB per launch", sum(v)/len(v), "launches", len(v), " x2 bytes =", 2*1024*sum(v)/len(v))
P
done
