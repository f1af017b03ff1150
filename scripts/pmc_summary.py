"""Summarise `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of scripts/pmc_op.py into HBM bytes per
launch of an op.   pmc_summary.py <op> <N> <fetch_dir> <write_dir> [exclude-kernel-substring ...]
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 this rocprofv3 reports exactly half of the bytes of a wide
coalesced read in FETCH_SIZE (MI355X_MICROARCH.md, HBM section): it is doubled here; WRITE_SIZE is taken
as reported (uncalibrated per the guide)."""
import csv, glob, json, os, sys
from collections import defaultdict


def collect(d, cnt, exclude):
    per = defaultdict(lambda: [set(), 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != cnt:
                continue
            k = row["Kernel_Name"]
            if any(e in k for e in exclude):
                continue
            per[k[:100]][0].add(row.get("Dispatch_Id"))
            per[k[:100]][1] += float(row["Counter_Value"])
    return {k: {"dispatches": len(v[0]), "sum_KiB": v[1]} for k, v in per.items()}


def main():
    op, n = sys.argv[1], int(sys.argv[2])
    exclude = sys.argv[5:]
    fetch = collect(sys.argv[3], "FETCH_SIZE", exclude)
    write = collect(sys.argv[4], "WRITE_SIZE", exclude)
    fk = sum(v["sum_KiB"] for v in fetch.values()) / n
    wk = sum(v["sum_KiB"] for v in write.values()) / n
    out = {"op": op, "launches": n, "FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk,
           "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024),
           "correction": "FETCH_SIZE x2 (gfx950: half of wide coalesced reads counted), WRITE_SIZE as reported",
           "kernels_fetch": fetch, "kernels_write": write}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
