#!/bin/bash
# SQ counters (two --pmc passes, no trace domains) of named kernels inside one bench workload:
#   gpurun -- 'bash scripts/sq_kernels.sh TAG retinanet_100k tk_gather tk_hist tk_pool1 tk_compact'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; REPO=$PWD; TAG=${1:-sq}; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; WL=$2; shift 2
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"
B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS"   # (r05: MFMA pipe busy cycles -- cycles, not quad-cycles)
cd /tmp
for P in A B C; do
  eval CNT=\$$P
  rm -rf /tmp/sq_$P; timeout 300 rocprofv3 --pmc $CNT --output-format csv -d /tmp/sq_$P -o p -- ${SQ_CMD:-python $REPO/bench.py --workload $WL --no-cpu-baseline --steps 3 --warmup 1} > $OUT/sq_$P.log 2>&1; echo "pass $P rc=$?"
done
python - "$OUT" "$@" <<'PY'
import csv, glob, sys, json, collections
out, names = sys.argv[1], sys.argv[2:]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for P in "ABC":
    for f in glob.glob("/tmp/sq_%s/**/*counter_collection.csv" % P, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            for n in names:
                if n in k:
                    res[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
json.dump(summ, open(out + "/sq_counters.json", "w"), indent=1)
for k, d in summ.items():
    print(k[:60], " ".join("%s=%.3g" % (c.replace("SQ_", ""), v) for c, v in sorted(d.items())))
PY
