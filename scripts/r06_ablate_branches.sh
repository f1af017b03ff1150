cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH=$PWD
for REP in 1 2; do for A in none anchors subsample crop anchors,crop; do
  D2AMD_BENCH_ABLATE=$A timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ablate $A', d['ms_per_step'])" || echo "ablate $A failed"
done; done
