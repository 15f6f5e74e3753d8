import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle_chain():
    """Builds (if needed) and loads the C restatement."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "_build", "libcoper_oracle.so")
    src = os.path.join(ROOT, "oracle", "coper_oracle_chain.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    from oracle import coper_oracle as O
    O.chain_lib()
    return O
