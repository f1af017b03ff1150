"""Print per-dispatch durations (us), in time order, of kernels whose name contains argv[2:] -- from a
rocprofv3 *_kernel_trace.csv (argv[1])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for nm in sys.argv[2:]:
    s = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if nm in r['Kernel_Name']]
    print(nm, len(s), [round(x, 1) for x in s[-10:]])
