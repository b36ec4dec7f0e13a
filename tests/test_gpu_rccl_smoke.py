"""RCCL (torch.distributed backend "nccl"), torch's HIP runtime and libgdhip.so in one process on the GPU box: a single-rank
communicator, all-gathers of GPU tensors around and during batched calls.  The multi-rank logic is covered by the gloo
tests (tests/test_parallel_gloo.py); multi-GPU runs are the driver's."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_and_the_library_share_a_process():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "nccl_smoke.py")], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0 and "nccl smoke ok" in out.stdout, out.stdout[-3000:]
