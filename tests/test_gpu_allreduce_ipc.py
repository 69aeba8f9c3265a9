"""The process-group constructor of the one-shot all-reduce: two PROCESSES (one GPU: the only topology a 1-GPU box offers)
exchange CUDA-IPC handles through a gloo group and reduce through each other's mapped buffers, eagerly and inside a hipGraph."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_processes_one_gpu_ipc_allreduce():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(HERE, "_allreduce_two_proc.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "rank 0: ok=True" in out and "rank 1: ok=True" in out, out[-3000:]
