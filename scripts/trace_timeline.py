"""Timeline of the last frame in a rocprofv3 --kernel-trace csv of scripts/prof_run.py: per dispatch its start
(us after the frame's first kernel), duration and the gap since the previous kernel ended.
usage: trace_timeline.py <dir> [frames_back]"""
import csv, glob, sys
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for f in sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    gen = [i for i, r in enumerate(rows) if "k_generate" in r["Kernel_Name"]]
    if len(gen) < back:
        continue
    lo = gen[-back]
    hi = gen[-back + 1] if back > 1 else len(rows)
    t0 = int(rows[lo]["Start_Timestamp"])
    prev_end = t0
    tot = 0.0
    print(f)
    for r in rows[lo:hi]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("nova_hip::", "")[:40]
        print("%9.1f us  +%7.1f us  gap %6.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
        prev_end = e
        tot += (e - s) / 1e3
    print("frame: %.1f us from first start to last end, %.1f us inside kernels" % ((prev_end - t0) / 1e3, tot))
