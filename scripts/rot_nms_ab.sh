#!/bin/bash
# rotated NMS: tests (both mask variants), then rrpn_micro with the compacted mask and with D2AMD_NMS_ROT_PLAIN=1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONPATH=$PWD; OUT=gpurun_out/${1:-rot_nms_ab}; mkdir -p $OUT
for P in 0 1; do
  if [ $P = 1 ]; then export D2AMD_NMS_ROT_PLAIN=1; else unset D2AMD_NMS_ROT_PLAIN; fi
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_tests.py tests/test_gpu_nms_runs.py -m gpu -q -p no:cacheprovider -k "rot or Rot or nms" 2>&1 | tail -2
done
for R in 1 2; do for P in 0 1; do
  if [ $P = 1 ]; then export D2AMD_NMS_ROT_PLAIN=1; else unset D2AMD_NMS_ROT_PLAIN; fi
  timeout 300 python bench.py --workload rrpn_micro --no-cpu-baseline > $OUT/rrpn_plain${P}_$R.json 2> $OUT/rrpn_plain${P}_$R.err
  python -c "import json; d=json.load(open('$OUT/rrpn_plain${P}_$R.json')); print('plain=$P', d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['ops'].items()}, d['roofline']['kernels_ms'], d['config']['nms_kept'])"
done; done
