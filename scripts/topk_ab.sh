#!/bin/bash
# Large-segment selection (csrc/topk.hip): tests, then per setting (name:VAR=VAL ...) the selection alone
# (topk_select_bench.py: wall time per call + rocprofv3 kernel stats of it) and the retinanet_100k bench line, on one box.
#   gpurun -- 'bash scripts/topk_ab.sh TAG new: legacy:D2AMD_TOPK_LEGACY=1'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-topk_ab}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; shift
timeout 900 python -m pytest tests/test_gpu_topk_large.py tests/test_gpu_dense.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for S in "$@"; do
  NAME=${S%%:*}; KV=${S#*:}
  for REP in 1 2; do
    env $KV timeout 300 python bench.py --workload retinanet_100k --no-cpu-baseline > $OUT/bench_${NAME}_$REP.json 2> $OUT/bench_${NAME}_$REP.err
    python -c "import json; d=json.load(open('$OUT/bench_${NAME}_$REP.json')); print('$NAME bench', d['ms_per_step'], d['value'])"
  done
  env $KV timeout 300 python scripts/topk_select_bench.py 2> /dev/null | tee $OUT/select_$NAME.json
  rm -rf /tmp/tk; (cd /tmp && env $KV timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk -o m -- python $REPO/scripts/topk_select_bench.py 20 > /dev/null 2>&1)
  f=$(find /tmp/tk -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/select_kernel_stats_$NAME.csv
  python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/select_kernel_stats_$NAME.csv")))
tk = [(r["Name"].split("(")[0].replace("d2amd::", "").replace("void ", ""), float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in rows if "d2amd" in r["Name"]]
print("  ", "$NAME", " ".join("%s %.1f(x%d)" % (n, a, c) for n, a, c in tk))
PY
done
