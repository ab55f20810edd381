"""GPU suite: the training path under DistributedDataParallel (trainer.py:19) — two ranks on the box's single GPU, gloo
collectives (`DIGAT_BENCH_TEST_SHARED_GPU`, the hook bench.py's two-rank test uses).  Each rank takes half a batch; the
averaged gradients must equal the single-process gradients of the whole batch (tools/ddp_check.py: 1e-6 + 1e-5 of each
tensor's largest gradient; dropout 0)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ddp_gradients_equal_single_process_gradients_of_the_union():
    env = dict(os.environ, DIGAT_BENCH_TEST_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29741", os.path.join("tools", "ddp_check.py")]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert res.returncode == 0 and lines, (res.stdout[-1500:], res.stderr[-2500:])
    out = json.loads(lines[-1])
    assert out["ok"] and out["world"] == 2 and out["params"] >= 40, out
