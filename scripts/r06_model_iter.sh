#!/bin/bash
# r06: whole-model iterations of the reference's Mask R-CNN, layer-level binding vs the fused callers (integrate.patch):
# s / iteration, host syncs per iteration, and the library's own kernel time per iteration (rocprofv3 kernel stats)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-r06_model_iter}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
for MODE in "" "--fused"; do
  N=layer; [ -n "$MODE" ] && N=fused
  timeout 600 python scripts/model_iteration_bench.py --amp --iters 20 --count-syncs $MODE 2> $OUT/iter_$N.err | tail -1 | tee $OUT/iter_$N.json
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$N -o p -- python $REPO/scripts/model_iteration_bench.py --amp --iters 10 --warmup 3 $MODE > $OUT/prof_$N.log 2>&1)
  F=$(find $OUT/prof_$N -name "*kernel_stats.csv" | head -1)
  python - "$F" $N <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "d2amd" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(sys.argv[2], "library kernels: %d kinds, %.3f ms per iteration (13 iterations)" % (len(rows), tot / 1e6 / 13))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("   %-100s %5s %8.3f ms" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6))
PY
  cp $F $OUT/${N}_kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof_$N
done
