"""CPU: bench.py's launch contract.  `--gpus N` must never report N GPUs unless N ranks are running: outside a launcher it
starts the ranks itself (torch.distributed.run on 127.0.0.1), inside one it refuses when WORLD_SIZE != N."""
import os
import subprocess
import sys
from unittest import mock

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2, r.stderr
    assert "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""       # no JSON line for GPUs that are not running


def test_self_launch_starts_one_rank_per_gpu():
    sys.path.insert(0, ROOT)
    import bench
    with mock.patch.object(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1"]):
        args = bench.parse_args()
        with mock.patch.object(bench.subprocess, "call", return_value=0) as call:
            assert bench.self_launch(args) == 0
    cmd = call.call_args[0][0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "2", "--warmup", "1"]
    assert call.call_args[1]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
