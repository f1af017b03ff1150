cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH=$PWD
for R in 1 2; do for M in none side all chained; do
  D2AMD_PREBIN=$M timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$M', d['ms_per_step'], d.get('ms_per_step_windows'))"
done; done
