"""RCCL on the GPU box.  The round's boxes have one GPU, so the neighbour exchange itself is covered by the gloo tests
(tests/test_cpu_parallel.py, tests/test_cpu_cli_fanout.py, the two-ranks-on-one-GPU CLI test); here: the RCCL backend loads
and accepts, at world size 1, every call the N > 1 path of bench.py and parallel.py makes."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_rccl_backend_single_rank(gpu):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_single_rank_child.py"), str(port)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
