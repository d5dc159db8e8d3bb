"""Per-dispatch durations from a rocprofv3 --kernel-trace csv (kernel name prefix, ms)."""
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    out = []
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nova_hip::", "")[:18]
        out.append("%s %.3f" % (name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    print(f)
    print(" | ".join(out[-40:]))
