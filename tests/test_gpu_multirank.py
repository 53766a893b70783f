"""SURVEY.md section 8(e) with more than one rank on real device state (needs an MI355X; one is enough).

A test box has ONE GPU and RCCL does not put two ranks of a communicator on one device, so the two-rank runs below go
through bench.py's ALTRO_BENCH_BACKEND=gloo hook: everything on the path is the real thing -- two processes, two
handles on the GPU, their own shards of the global batch, the device-side statistics reduction of the C ABI -- and
only the two tiny all-reduces travel over gloo instead of RCCL.  (World-size-1 RCCL is tests/test_gpu_stats.py and
tests/rccl_rank_check.py; the 1/2/4/8-GPU RCCL curve is the driver's.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ, ALTRO_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    return env


@pytest.mark.timeout(600)
def test_two_ranks_on_real_handles_reduce_like_one_process():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_port()), os.path.join(ROOT, "tests", "multirank_check.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "multirank check OK: world 2" in r.stdout


def _bench(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=_env(), capture_output=True,
                       text=True, timeout=560)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_bench_honours_gpus_2_weak_c1():
    """`python bench.py --gpus 2` (no launcher around it) runs two ranks and says so."""
    out = _bench(["--gpus", "2", "--batch", "256", "--horizon", "32", "--steps", "3", "--warmup", "1",
                  "--repeat-seconds", "0", "--no-cpu-baseline"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 512 and out["config"]["batch_per_gpu"] == 256
    st = out["config"]["stats"]
    assert "world 2" in st["reduced_by"]
    assert st["problems"] == 512 and st["converged"] == 512 and st["cholesky_failures"] == 0 and st["non_finite"] == 0


@pytest.mark.timeout(600)
def test_bench_c3_shards_the_global_batch():
    """configs[3] names a node-wide batch: --config c3 splits --global-batch over the ranks (ragged here)."""
    out = _bench(["--gpus", "2", "--config", "c3", "--global-batch", "1001", "--steps", "3", "--warmup", "1",
                  "--repeat-seconds", "0"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["global_batch"] == 1001 and out["config"]["batch_per_gpu"] == 501
    st = out["config"]["stats"]
    assert "world 2" in st["reduced_by"] and st["problems"] == 1001
    assert out["value"] == pytest.approx(1001 * 3 / (out["ms_per_step"] * 3e-3), rel=1e-9)
