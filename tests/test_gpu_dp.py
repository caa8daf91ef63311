"""Data-parallel train step on the GPU: two ranks sharing cuda:0 over gloo (a one-GPU box cannot host two RCCL ranks; the
collective is the only thing that differs from the 8-GPU run, and ``dp.allreduce_mean_`` has a gloo branch for exactly this).
The launched script (tools/rehearse_dp_gpu.py) asserts that the replicas stay bit-identical over two steps on two side
streams each, and that the mean of the per-rank losses equals a single process on the global batch."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_train_step_on_one_gpu():
    """The ranks are started by the GPU-clean helper tests/conftest.py spawned at session start (a process that has initialised the
    GPU may not exec on this pool), so the test does not depend on running before the other GPU tests; a free port is picked."""
    import socket

    from conftest import launch_clean
    if torch.cuda.device_count() < 1:
        pytest.skip("no GPU")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "rehearse_dp_gpu.py")]
    out = launch_clean(cmd, env=env, cwd=ROOT, timeout=600)  # raises (= the test FAILS) when there is no clean launcher
    print(out["stdout"][-2000:])
    assert out["returncode"] == 0, out["stderr"][-3000:]
    assert "[dp rehearsal] ok" in out["stdout"]
