"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, sum of each counter
over dispatches and the dispatch count."""
import csv, glob, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:40]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in agg:
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-34s %18.0f   (%d dispatches, %.1f per dispatch)" % (c, v, cnt[k][c], v / cnt[k][c]))
