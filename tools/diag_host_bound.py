"""Diagnostic (GPU box): is the training step host-bound?  Time to ENQUEUE K steps vs time until the GPU has finished them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-style-transfer-gan_amd")]
import torch, enhanced_train
torch.manual_seed(0)
m = enhanced_train.EnhancedCycleGAN(channels=16, num_transformer_blocks=0, device=torch.device("cuda", 0))
a = (torch.rand((32, 3, 256, 256)) * 2 - 1).cuda(); b = (torch.rand((32, 3, 256, 256)) * 2 - 1).cuda()
for _ in range(3): m.train_step_async(a, b)
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K): m.train_step_async(a, b)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / K:.1f} ms/step   until GPU done {1e3 * (t2 - t0) / K:.1f} ms/step   (host idle at the end {1e3 * (t2 - t1):.1f} ms)")
