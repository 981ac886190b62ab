"""N > 1 on the GPU box: two ranks share the one GPU (gloo carries the collective; on the 8-GPU node the same code runs over
RCCL) and must reproduce the single-pipeline visible set through the stream-ordered slab all-gather."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_slab_allgather_two_ranks():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(HERE, "slab_allgather_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "OK slab all-gather" in out.stdout
