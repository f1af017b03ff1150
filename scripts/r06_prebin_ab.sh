cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD
for REP in 1 2 3; do for M in 0 1 2; do
  D2AMD_BENCH_PREBIN=$M timeout 300 python bench.py --no-cpu-baseline --no-extra-workloads 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prebin $M', d['ms_per_step'], d['roofline']['kernels_ms'])"
done; done
