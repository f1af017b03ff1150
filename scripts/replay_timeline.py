"""Timeline of ONE replayed step of bench.py's connected graph out of a rocprofv3 kernel trace (csv): start offset,
duration, HSA queue, name of every kernel; `|` = starts while an earlier kernel is still running.
usage: replay_timeline.py <kernel_trace.csv> [step index from the end of the replayed region, default 3]"""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
# a step starts with the proposal selection's fused select; the anchor sampler launches the same kernel later in the step
idx = [i for i, e in enumerate(ev) if "tk_fused_kernel" in e[2]]
st = np.array([ev[i][0] for i in idx])
d = np.diff(st) / 1e3
per = [j for j in range(len(d) - 2) if d[j] + d[j + 1] < 700 and abs((d[j] + d[j + 1]) - (d[j + 1] + d[j + 2])) < 30]
sel = [j for j in per if d[j] < d[j + 1]]  # selection first, sampler second
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
j = sel[-k]
a, b = idx[j], idx[j + 2]
while a > 0 and ev[a][0] - ev[a - 1][1] < 3000 and ev[a - 1][1] > ev[idx[j - 1]][1]:  # the clears / key launches in front
    a -= 1
    if "pool_bwd" in ev[a][2]:
        a += 1
        break
t0, end = ev[a][0], 0
for s, e, n, q in ev[a:b]:
    if "pool_bwd" in n and s > ev[idx[j + 1]][0]:
        pass
    n = n.replace("d2amd::", "").replace("void ", "")
    mark = " " if s >= end else "|"
    print(f"{(s - t0) / 1e3:8.1f} {mark} {(e - s) / 1e3:6.1f} q{q} {n[:96]}")
    end = max(end, e)
print(f"step period {(ev[idx[j + 2]][0] - ev[idx[j]][0]) / 1e3:.1f} us")
