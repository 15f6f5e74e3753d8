"""GPU: a short run of the randomised soak (tests/soak.py): model shapes, table scales, batch sizes around every tile boundary,
skewed and duplicated queries, empty to very long filter rows -- the fused pass's ranks and tie counts must equal the fp32 chain's
on the pass's own embeddings for every query of every case, the two-call path must agree bit for bit, the band audit stays
below 0.5.  (The long runs behind DESIGN.md's claim: `python tests/soak.py 400 <seed>`, seeds 1 - 3 at the end of round 4.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_soak_short(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak.py"), "14", str(seed)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "all ranks == the fp32 chain's" in out.stdout


def test_eval_stream_example():
    """examples/eval_stream.py: passes whose batches arrive beside the previous pass's encoder (coper_stage_ids_next) and whose ranks
    leave beside the next pass's first launch (coper_post_i32_next) -- the script itself checks every pass against a plain one."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "eval_stream.py"), "--batches", "5", "--queries", "6000"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "5 passes of 6000 queries" in out.stdout
