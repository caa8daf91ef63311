"""Diagnostic (GPU box): do the two generators' forward+backward overlap usefully on two HIP streams?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-style-transfer-gan_amd")]
import torch
import enhanced_generator as eg
dev = "cuda:0"
torch.manual_seed(0)
G1 = eg.EnhancedGenerator(16, 0).to(dev); G2 = eg.EnhancedGenerator(16, 0).to(dev)
x = torch.rand((64, 3, 256, 256), device=dev) * 2 - 1
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
main = torch.cuda.current_stream()

def seq():
    y1 = G1(x); y2 = G2(x)
    (y1.abs().mean() + y2.abs().mean()).backward()

def par():
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1):
        y1 = G1(x); l1 = y1.abs().mean()
    with torch.cuda.stream(s2):
        y2 = G2(x); l2 = y2.abs().mean()
    main.wait_stream(s1); main.wait_stream(s2)
    l1.record_stream(main); l2.record_stream(main)
    (l1 + l2).backward()

for name, fn in (("sequential", seq), ("two streams", par), ("sequential", seq), ("two streams", par)):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(f"{name:12s} {(time.perf_counter() - t0) / 5 * 1e3:8.2f} ms per fwd+bwd of both generators (N64)")
