"""Summarise a `rocprofv3 --pmc <COUNTER>` run: per kernel name, dispatch count and mean counter
value.  Usage: pmc_summary.py <dir> <COUNTER>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, cnt = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: [set(), 0.0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != cnt:
                    continue
                k = row["Kernel_Name"][:120]
                a = acc[k]
                a[0].add(row.get("Dispatch_Id", str(len(a[0]))))
                a[1] += float(row["Counter_Value"])
    # rocprofv3 emits one row per (dispatch, counter, dimension instance): sum instances per dispatch
    out = {k: {"dispatches": len(v[0]), "sum": v[1], "mean_per_dispatch": v[1] / max(len(v[0]), 1)}
           for k, v in acc.items()}
    print(json.dumps({"counter": cnt, "kernels": out}))


if __name__ == "__main__":
    main()
