"""CPU, world size 2, gloo: the data-parallel glue (mstg_hip/dp.py) -- equal image shards + ONE averaged flat gradient
buffer per optimizer reproduce the global-batch gradient of the reference's losses."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": str(rank)})
    torch.set_num_threads(2)
    from mstg_hip import dp
    from oracle import restatement as R
    dp.init_from_env("gloo")
    assert dp.world_size() == world and dp.rank() == rank
    C, shape = 8, (4, 3, 32, 32)
    g_sd = {k: v.requires_grad_(True) for k, v in R.make_state_dict(R.generator_spec(C), 5).items()}
    d_sd = R.make_state_dict(R.discriminator_spec(C), 6)
    keys = [k for k in g_sd if not k.startswith("style_encoder")]
    real = R.make_input(shape, 7)

    def flat_grad(batch):
        for k in d_sd:  # every rank runs the same number of power iterations from the same u, v
            pass
        y = R.generator_forward(g_sd, batch)
        score, struct = R.discriminator_forward({k: v.clone() for k, v in d_sd.items()}, y, train=True)
        loss = R.mse(score, 1.0) + 10.0 * R.l1(y, batch) + 0.5 * struct.abs().mean()
        return torch.cat([g.flatten() for g in torch.autograd.grad(loss, [g_sd[k] for k in keys])]), float(loss)

    local = dp.shard(real)
    assert local.shape[0] == shape[0] // world and torch.equal(local, real[rank * 2:(rank + 1) * 2])
    g_local, _ = flat_grad(local)
    g_avg = dp.allreduce_mean_(g_local.clone())
    g_global, _ = flat_grad(real)
    rel = float((g_avg - g_global).norm() / g_global.norm())
    # broadcast: rank 1 starts from garbage and must end with rank 0's buffer
    buf = torch.full((1000,), float(rank + 1))
    dp.broadcast_(buf, src=0)
    q.put((rank, rel, float(buf.mean())))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradient_equivalence_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rel, mean in out:
        assert rel <= 1e-4, (rank, rel)          # mean of shard gradients == global-batch gradient (fp32 reduction order)
        assert mean == 1.0


def test_shard_rejects_ragged_batch_single_process():
    from mstg_hip import dp
    assert dp.world_size() == 1 and dp.rank() == 0
    x = torch.zeros(5, 3)
    assert dp.shard(x) is x and dp.allreduce_mean_(x) is x
