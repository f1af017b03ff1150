cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_pooler.py tests/test_gpu_pooler_pair.py tests/test_gpu_reference_callers.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], 'nchw_drop_in', d.get('nchw_drop_in_ms'), 'clustered', d.get('clustered_rois_ms'))"; done
