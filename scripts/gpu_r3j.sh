#!/bin/bash
# round 3, visit j: kernel stats + trace of the connected one-graph step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; OUT=$REPO/gpurun_out/r3l; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_label_sample.py tests/test_gpu_connected_step.py tests/test_gpu_graph.py tests/test_gpu_rpn.py -q -m gpu 2>&1 | tail -5
timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads > $OUT/bench.json 2>$OUT/bench.err; python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['roofline']['kernels_ms'])"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-workloads > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
for r in rows[:45]: print(f"{r['Name'][:110]:110s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}")
PY
# one replayed step's kernel timeline (last graph replay in the trace)
python - <<PY
import csv,glob
f=glob.glob("$OUT/prof/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find the timed region: take the last 400 kernels before the tail and print one step's worth from a tk_fused to the next
idx=[i for i,r in enumerate(rows) if 'tk_fused' in r['Kernel_Name']]
a=idx[14]; b=idx[15]
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} {(int(r['End_Timestamp'])-t0)/1e3:8.1f}  q{r.get('Queue_Id','?'):>3s} {r['Kernel_Name'][:90]}")
PY
find $OUT/prof -type f -name "*kernel_trace.csv" -size +4M -delete
