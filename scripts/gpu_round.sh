#!/bin/bash
# One GPU visit for the round's evidence: full GPU suite, the three bench workloads (+ NCHW), rocprofv3 kernel stats of
# each.  Everything lands in gpurun_out/$1/ (copy what is to be judged into profiles/).
O=gpurun_out/${1:-round}; mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
for wl in maskrcnn_train retinanet_100k dcn_r50; do
  timeout 900 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "rc=$? $wl"
done
timeout 600 python bench.py --layout nchw --no-cpu-baseline > $O/bench_maskrcnn_train_nchw.json 2> $O/bench_nchw.err
timeout 600 python bench.py --no-overlap --no-cpu-baseline > $O/bench_maskrcnn_train_one_stream.json 2> $O/bench_one_stream.err
timeout 600 python bench.py --no-graph --no-cpu-baseline > $O/bench_maskrcnn_train_eager.json 2> $O/bench_eager.err
export TMPDIR=/tmp
for wl in maskrcnn_train retinanet_100k dcn_r50; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$wl -o $wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_$wl.log 2>&1)
  f=$(find $O/prof_$wl -name "*kernel_stats.csv" | head -1); cp "$f" $O/${wl}_kernel_stats.csv
  rm -rf $O/prof_$wl
done
python - <<PY
import json
for wl in ("maskrcnn_train", "retinanet_100k", "dcn_r50", "maskrcnn_train_nchw", "maskrcnn_train_one_stream", "maskrcnn_train_eager"):
    try:
        d = json.load(open("$O/bench_%s.json" % wl))
        r = d.get("roofline", {})
        print(wl, d["value"], d["unit"], d["ms_per_step"], "ms/step; roofline", r.get("kernel", "")[:40], r.get("frac"), "cpu", d.get("cpu_baseline", {}).get("value"))
    except Exception as e:
        print(wl, "failed", e)
PY
