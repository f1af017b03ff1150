"""The multi-GPU plumbing of bench.py on the ONE GPU a test box has: `--force-dist` creates the RCCL process group at
world size 1 (ProcessGroupNCCL init, the gradient buckets' asynchronous all-reduce between the step's two HIP graphs,
barrier-bracketed MAX timing) -- what `--gpus 8` does per rank.  The 2-rank logic (shards, bucket averaging, launcher)
is covered on CPU with gloo in tests/test_sharding_gloo.py; the scaling curve is the driver's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_force_dist_initialises_rccl_and_reduces():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.pop("MASTER_PORT", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "5",
                        "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    a = d["allreduce"]
    assert a["backend"] == "nccl" and a["world"] == 1 and a["values_ok"] is True
    assert a["buckets"] == 3 and 80 < a["wire_MB"] < 100          # 44.1 M parameters in bf16
    assert a["alone_ms"] > 0 and a["value_data_path"] >= d["value"] * 0.95
    assert "2 HIP graphs per step (forward | backward" in d["config"]["launch"]
    assert "gradient all-reduce of 44,120,816 parameters" in d["config"]["parallelism"]
