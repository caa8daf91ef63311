"""Diagnostic (GPU box): train_step with gradient checkpointing on/off and with one/two streams gives the same numbers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-style-transfer-gan_amd")]
import torch, enhanced_train
def run(ckpt, streams):
    torch.manual_seed(3)
    m = enhanced_train.EnhancedCycleGAN(channels=16, num_transformer_blocks=0, device=torch.device("cuda", 0), gradient_checkpointing=ckpt)
    m.two_streams = streams
    g = torch.Generator().manual_seed(5)
    a = (torch.rand((4, 3, 64, 64), generator=g) * 2 - 1).cuda(); b = (torch.rand((4, 3, 64, 64), generator=g) * 2 - 1).cuda()
    out = [m.train_step(a, b) for _ in range(3)]
    return out, m.g_optimizer.flat.clone(), m.d_optimizer.flat.clone()
ref = run(False, False)
for ckpt, streams in ((False, True), (True, False), (True, True)):
    o = run(ckpt, streams)
    same_loss = all(o[0][i][k] == ref[0][i][k] for i in range(3) for k in ref[0][i])
    print(f"checkpointing={ckpt} two_streams={streams}: losses bit-identical to (False, False): {same_loss}; "
          f"G params identical: {torch.equal(o[1], ref[1])}; D params identical: {torch.equal(o[2], ref[2])}")
