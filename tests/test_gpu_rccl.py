"""RCCL on the GPU box: the one collective of the path, the weight broadcast, executed for real.
A one-GPU box cannot run two ranks on RCCL (one rank per device), so the world has one rank -- which still loads
librccl, builds a communicator and launches its broadcast / all-reduce kernels on the real 3.4 MB blob.  The 1 -> 8
GPU curve is the driver's to measure; this is the part of it a one-GPU box can execute.  The child is a fresh
process (never a re-exec of one that touched the GPU); a non-zero exit code fails the test."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_rccl_broadcast_of_the_weight_blob_in_a_fresh_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "rccl_child.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    res = subprocess.run([sys.executable, child, str(port)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "RCCL_OK world=1 blob_floats=856788" in res.stdout
