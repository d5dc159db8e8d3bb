"""Reads a rocprofv3 kernel-trace CSV and reports, over the span of the LAST `samples` samples' kernels,
how much of the wall time had 1, 2, 3+ kernels in flight and the summed kernel time per wall time.
usage: trace_overlap.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ev.sort()
# the second half of the kernels = the timed call (the script makes two identical calls)
half = ev[len(ev) // 2:]
t0, t1 = half[0][0], max(e[1] for e in half)
pts = []
for s, e, _ in half:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
depth, last, hist = 0, t0, {}
for t, d in pts:
    hist[depth] = hist.get(depth, 0) + (t - last)
    last = t
    depth += d
wall = t1 - t0
busy = sum(e - s for s, e, _ in half)
print(f"kernels {len(half)}, wall {wall / 1e6:.3f} ms, summed kernel time {busy / 1e6:.3f} ms ({busy / wall:.2f}x)")
for k in sorted(hist):
    print(f"  {k} kernel(s) in flight: {hist[k] / 1e6:8.3f} ms  {100.0 * hist[k] / wall:5.1f} %")
