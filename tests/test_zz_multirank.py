"""Multi-rank rehearsal on one GPU (collected last: a failure here must not hide the parity tests under -x)."""
import pytest

pytestmark = pytest.mark.gpu


def test_two_rank_bench_on_one_gpu():
    """bench.py end to end with two ranks sharing the GPU (GAN_AMD_ONE_GPU=1: gloo, host-staged collectives):
    broadcast, bucketed all-reduce from the hooks, graph capture agreed across ranks, flat all-reduces between
    the graph segments, the event-bracketed iteration on every rank, MAX over ranks.  A collective that only one
    rank issues shows up here as a hang (bounded by the watchdog and the timeout)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GAN_AMD_ONE_GPU="1", GAN_AMD_BENCH_WATCHDOG="100")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no_cpu",
           "--no_micro", "--img_size", "64", "--base_dim", "64", "--global_batch", "16"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=150)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["per_gpu_batch"] == 8
    assert line["config"]["hip_graph"] and line["config"]["losses_finite"]
    assert line["roofline"] is not None and line["value"] > 0


def test_two_rank_trainer_matches_global_batch():
    """Two ranks, one GPU: the real Trainer on interleaved shards of one global batch (global random draws, sliced)
    must hand the optimizers the global-batch gradients -- D (hinge + R1, MinibatchStdDev included) and G -- on the
    hook path (bucketed all-reduces) and on the cut path the HIP graphs capture (tests/dp_parity_worker.py)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GAN_AMD_ONE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29573", os.path.join(root, "tests", "dp_parity_worker.py")]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=150)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "DP_PARITY_OK" in out.stdout
