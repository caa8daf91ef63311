"""Data-parallel train step on the GPU: two ranks sharing cuda:0 over gloo (a one-GPU box cannot host two RCCL ranks; the
collective is the only thing that differs from the 8-GPU run, and ``dp.allreduce_mean_`` has a gloo branch for exactly this).
The launched script (tools/rehearse_dp_gpu.py) asserts that the replicas stay bit-identical over two steps on two side
streams each, and that the mean of the per-rank losses equals a single process on the global batch."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_train_step_on_one_gpu():
    # device_count() does not initialise the GPU; a child may only be exec'ed from a process that has not touched it yet
    # (this module sorts first among the gpu-marked ones, so in the plain `pytest -m gpu` order that holds)
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU")
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: launching ranks from it is not allowed on the pool")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "tools", "rehearse_dp_gpu.py")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    print(out.stdout[-2000:])
    assert out.returncode == 0, out.stderr[-3000:]
    assert "[dp rehearsal] ok" in out.stdout
