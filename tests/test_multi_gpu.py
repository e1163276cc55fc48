"""N > 1 on hardware: runs only where at least two GPUs are visible — the GPU box of the
development pool has one, the driver's 8-GPU node has eight — and skips with that reason otherwise.  The same code
paths run at world size 2 on CPU (gloo + the emulator build) in tests/test_sharding.py and tests/test_bench_path.py."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gpus():
    import torch

    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_two_ranks_broadcast_and_shard():
    n = _gpus()
    if n < 2:
        pytest.skip(f"{n} GPU visible: the two-rank RCCL check needs two (it runs on the driver's multi-GPU node)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(REPO / "tests" / "multi_gpu_check.py")]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert p.stdout.count("MULTI_GPU_CHECK") == 2 and p.stdout.count(" OK") >= 2


def test_bench_runs_on_two_gpus():
    n = _gpus()
    if n < 2:
        pytest.skip(f"{n} GPU visible: `bench.py --gpus 2` needs two (the driver's scaling run launches it on its 8-GPU node)")
    cmd = [sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-half-mode",
           "--no-config4", "--no-config5", "--config3-utterances", "16"]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert "nccl" in line["process_group"] and line["weight_broadcast_seconds"] is not None
    assert len(line["per_rank"]["utterances_per_sec"]) == 2
    assert sum(line["config3"]["shard_sizes"]) == 16
