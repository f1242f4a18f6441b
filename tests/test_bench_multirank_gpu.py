"""The N > 1 path of bench.py (one process per rank, FlatGradReducer buckets overlapped with the engine's backward, fused Adam
with the 1/world averaging folded in) rehearsed with TWO ranks on the one GPU of the test box: gloo instead of RCCL (which
refuses two ranks on one device), both ranks on device 0.  Checked: the job prints one well-formed line for n_gpus = 2, the loss
is finite, and the replicas' weights are BIT-identical after the steps although every rank saw different data (`replica_drift`
0.0: a broken or skipped exchange makes them diverge at the first step).  Both wire formats and both scaling modes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra,port,exchange", [(["--grad-dtype", "bf16"], 29641, True),
                                                 (["--grad-dtype", "fp32", "--scaling", "strong", "--batch", "4"], 29642, True),
                                                 (["--grad-dtype", "bf16"], 29643, False)])        # negative control: no exchange
def test_bench_two_ranks_on_one_gpu(extra, port, exchange):
    env = dict(os.environ, MDE_BENCH_DEVICE="0", MDE_DIST_BACKEND="gloo", MDE_DP_BUCKET_MB="64", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if not exchange:
        env["MDE_DP_DISABLE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-launch-timing"] + (extra if "--batch" in extra else extra + ["--batch", "2"])
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["global_batch"] == 4 and d["config"]["per_gpu_batch"] == 2
    assert d["scaling"] == ("strong" if "strong" in extra else "weak")
    assert d["config"]["final_loss"] == d["config"]["final_loss"] and d["value"] > 0
    if exchange:
        assert d["config"]["replica_drift"] == 0.0, d["config"]
    else:
        assert d["config"]["replica_drift"] > 0.0, "the drift check cannot tell a working exchange from none"
