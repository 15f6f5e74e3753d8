"""GPU: the multi-GPU exchange code on a real RCCL communicator (one rank: the test box has one GPU).  The child is a
fresh process (never an exec from a process that touched the GPU): tests/rccl_child.py."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_sharded_evaluators_on_a_real_rccl_communicator():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py"), str(_free_port())], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "RCCL_CHILD_OK backend=nccl world=1" in out.stdout


def test_bench_entity_mode_through_rccl_with_one_rank():
    """bench.py's entity-sharded path with the process group forced on (COPER_BENCH_FORCE_DIST): the same code the
    driver's N > 1 runs take, RCCL collectives included, on a reduced entity count."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0", COPER_BENCH_FORCE_DIST="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "wn18rr_cpg", "--mode", "entity", "--topk", "10",
                          "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert lines, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads(lines[-1])
    assert line["config"]["parallelism"].startswith("entity-sharded x1") and line["value"] > 0
