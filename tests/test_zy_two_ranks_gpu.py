"""The sharded code paths with the real kernels: two ranks share the test box's GPU and exchange
data over gloo (see tests/dist_gpu_worker.py).  The RCCL transport itself needs two GPUs and is
exercised by bench.py on the multi-GPU node."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_share_one_gpu_over_gloo():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "dist_gpu_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0 and "DIST_GPU_OK" in out.stdout, (out.stdout + out.stderr)[-3000:]
