"""Rehearsal (one-GPU box): the data-parallel train step with 2 ranks sharing cuda:0 over gloo (RCCL needs one GPU per rank).
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/rehearse_dp_gpu.py
Checks: replicas stay bit-identical after two steps, and step-1 losses of the 2 x 2-pair run match a single process on the
4-pair batch (mean losses + averaged gradients = global-batch gradients)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-style-transfer-gan_amd")]
import torch, torch.distributed as dist
from mstg_hip import dp
import enhanced_train

dp.init_from_env("gloo")
rank, ws = dp.rank(), dp.world_size()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)

def build():
    torch.manual_seed(42)
    m = enhanced_train.EnhancedCycleGAN(channels=8, num_transformer_blocks=0, device=dev)
    m.sync_replicas()
    return m

g = torch.Generator().manual_seed(7)
A = torch.rand((2 * ws, 3, 64, 64), generator=g) * 2 - 1
B = torch.rand((2 * ws, 3, 64, 64), generator=g) * 2 - 1
m = build()
a, b = dp.shard(A).to(dev), dp.shard(B).to(dev)
l0 = m.train_step(a, b)
l1 = m.train_step(a, b)
# replicas identical?
chk = torch.stack([m.g_optimizer.flat.double().sum(), m.d_optimizer.flat.double().sum(), m.g_optimizer.flat.double().abs().sum()]).cpu()
allc = [torch.zeros_like(chk) for _ in range(ws)]
dist.all_gather(allc, chk)
same = all(torch.equal(allc[0], c) for c in allc)
# mean of the per-rank step-0 losses == global-batch loss of a single process
t = torch.tensor([l0[k] for k in sorted(l0)], dtype=torch.float64)
dist.all_reduce(t)
t /= ws
if rank == 0:
    print(f"[dp rehearsal] world {ws}: replicas identical after 2 steps: {same}")
    # single-process reference on the same device (no collectives: world_size() still 2, so bypass dp by monkeypatching)
    dp.allreduce_mean_ = lambda flat: flat
    dp.broadcast_ = lambda flat, src=0: flat
    torch.manual_seed(42)
    ref = enhanced_train.EnhancedCycleGAN(channels=8, num_transformer_blocks=0, device=dev)
    r0 = ref.train_step(A.to(dev), B.to(dev))
    tr = torch.tensor([r0[k] for k in sorted(r0)], dtype=torch.float64)
    err = ((t - tr).abs() / tr.abs().clamp_min(1e-12)).max().item()
    print(f"[dp rehearsal] step-0 losses, mean over ranks vs single process on the global batch: max rel diff {err:.2e}")
    assert same and err < 1e-5, (same, err)
    print("[dp rehearsal] ok")
dist.barrier()
dist.destroy_process_group()
