"""One step of a rocprofv3 kernel trace (csv) as a timeline: start offset, duration, name of every kernel.
usage: trace_timeline.py <kernel_trace.csv> [step index from the end, default 2]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
starts = [i for i, e in enumerate(ev) if "tk_fused_kernel" in e[2] or "tk_hist_kernel<0>" in e[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = starts[-k - 1], starts[-k]
a = max(0, a - 3)  # (the clears in front of the selection)
t0 = ev[a][0]
end = 0
for s, e, n in ev[a:b]:
    n = n.replace("d2amd::", "").replace("void ", "")
    mark = " " if s >= end else "|"  # | = starts while an earlier kernel is still running
    print(f"{(s - t0) / 1e3:8.1f} {mark} {(e - s) / 1e3:6.1f}  {n[:100]}")
    end = max(end, e)
print(f"step span {(ev[b][0] - t0) / 1e3:.1f} us")
