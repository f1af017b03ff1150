"""GPU busy / idle analysis of a rocprofv3 kernel trace (csv): per step of the timed region, how much of the wall time
has no kernel running.  usage: trace_gaps.py <kernel_trace.csv> [steps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steps are delimited by the fused top-k launch (first kernel family of a step)
starts = [i for i, e in enumerate(ev) if "tk_fused_kernel" in e[2]]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sel = starts[-nsteps - 1:]
tot_span = tot_busy = 0
gaps = []
for a, b in zip(sel[:-1], sel[1:]):
    seg = ev[a:b]
    t0, t1 = seg[0][0], ev[b][0]
    busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
    for s, e, n in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, prev, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        prev = n
    busy += cur_e - cur_s
    gaps.append((t1 - cur_e, seg[-1][2], "next step"))
    tot_span += t1 - t0
    tot_busy += busy
n = len(sel) - 1
print(f"{n} steps: span {tot_span / n / 1e3:.1f} us/step, GPU busy (union of kernels) {tot_busy / n / 1e3:.1f} us/step, idle {(tot_span - tot_busy) / n / 1e3:.1f} us/step")
agg = {}
for g, a, b in gaps:
    k = (a[:40], b[:40])
    agg.setdefault(k, [0, 0])
    agg[k][0] += g
    agg[k][1] += 1
for k, (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {g / n / 1e3:7.1f} us/step  ({c / n:.1f}x)  after {k[0]:40s} before {k[1]}")
