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


@pytest.mark.parametrize("dtype,algo,port", [("fp32", "allreduce", 29651), ("bf16", "allreduce", 29652),
                                             ("fp32", "rs_ag", 29653), ("bf16", "rs_ag", 29654)])
def test_bench_one_rank_over_rccl(dtype, algo, port):
    """The RCCL path itself (backend `nccl`: exchange stream, events, async collectives, wire buffers), which the two-rank
    rehearsal above cannot reach on one GPU: ONE rank launched the way the driver launches N, collectives forced although the
    world is 1 (a sum over one rank is the identity), both wire formats and both algorithms.  The loss must come out as in a
    run without any process group (fp32 wire: the same arithmetic; bf16 wire: gradients rounded once to 8 bits)."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "2", "--no-cpu-baseline",
            "--no-launch-timing"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r0 = subprocess.run(base, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r0.returncode == 0, r0.stderr[-3000:]
    ref = json.loads([l for l in r0.stdout.splitlines() if l.startswith("{")][0])
    env.update(MDE_DP_FORCE="1", MDE_DP_ALGO=algo, MDE_DP_BUCKET_MB="64")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + base[1:] + ["--gpus", "1", "--grad-dtype", dtype]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["replica_drift"] == 0.0
    a, b = d["config"]["final_loss"], ref["config"]["final_loss"]
    assert a == a and abs(a - b) <= (5e-3 if dtype == "fp32" else 2e-2) * abs(b), (a, b)


@pytest.mark.parametrize("config,act", [("bts", "bf16"), ("midas", "bf16"), ("vnl", "bf16"), ("vnl", "fp16")])
def test_bench_other_configurations_print_the_contract_line(config, act):
    """`bench.py --config` runs BASELINE.json configurations 3 / 4 / 5 through the same contract (here at batch 2) -- configuration
    5 also on the fp16 storage build, the precision BASELINE.json names for it (MDE_ACT_DTYPE=fp16; the line's dtype says so)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "2", "--warmup", "1", "--batch", "2"]
    env = dict(os.environ, MDE_ACT_DTYPE=act)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(r.stdout.strip().splitlines()) == 1, "bench.py must print exactly one line on stdout: %r" % r.stdout[:300]
    d = json.loads(r.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["config"]["per_gpu_batch"] == 2 and d["dtype"] == act
    assert d["config"]["final_loss"] == d["config"]["final_loss"]
    assert d["roofline"]["launches"] > 0 and 0.0 < d["roofline"]["frac"] < 1.0 and config.upper()[:3] in d["config"]["workload"].upper()


@pytest.mark.parametrize("config,port", [("midas", 29661), ("vnl", 29662)])
def test_bench_tape_config_one_rank_over_rccl_overlaps_the_exchange(config, port):
    """The tape networks' gradient exchange (BASELINE configurations 4 / 5, the 8-GPU ones) goes out DURING backward: one rank
    over RCCL with the collectives forced (as test_bench_one_rank_over_rccl), small buckets; the line says how many buckets
    were issued before backward had finished (all but the head of the buffer), replicas do not drift, and the loss equals a run
    without a process group."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "3", "--warmup", "1", "--batch", "2",
            "--no-launch-timing"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r0 = subprocess.run(base, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r0.returncode == 0, r0.stderr[-3000:]
    ref = json.loads([l for l in r0.stdout.splitlines() if l.startswith("{")][0])
    env.update(MDE_DP_FORCE="1", MDE_DP_BUCKET_MB="32")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + base[1:] + ["--gpus", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    c = d["config"]
    assert "overlapped with backward" in c["workload"] and c["replica_drift"] == 0.0
    assert c["exchange_buckets"] >= 3 and c["buckets_issued_before_backward_ended"] >= c["exchange_buckets"] - 2, c
    a, b = c["final_loss"], ref["config"]["final_loss"]
    assert a == a and abs(a - b) <= 5e-3 * abs(b), (a, b)
