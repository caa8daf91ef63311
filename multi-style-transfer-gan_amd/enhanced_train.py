"""Drop-in for the reference's ``enhanced_train.EnhancedCycleGAN`` (enhanced_train.py:13-152) on MI355X kernels.

Same surface: ``.G_AB .G_BA .D_A .D_B .g_optimizer .d_optimizer .criterion_* .lambda_*``,
``train_step(real_A, real_B) -> {'d_loss','g_loss','cycle_loss','identity_loss','structure_loss'}`` (python
floats) and ``save_models(save_dir, epoch)`` writing the reference's three checkpoint files.

Differences, all deliberate:
  * arithmetic is fp32 on the HIP kernels -- the parity target is the reference's CPU path, where
    ``torch.cuda.amp.autocast`` / ``GradScaler`` disable themselves; ``.scaler`` is a no-op stand-in;
  * the two Adam optimizers are ``FlatAdam`` (one fused launch, one contiguous gradient buffer each), and with
    ``torch.distributed`` initialised each gradient buffer is averaged over ranks by one collective before its step;
  * discriminator weight gradients are not computed during the generator phase (the reference computes and then
    discards them at the next ``zero_grad``, enhanced_train.py:67,121);
  * ``train_step_async`` returns the five losses as one device tensor with no host sync (the reference's
    ``train_step`` does five ``.item()`` syncs, :125-131); ``train_step`` wraps it with a single sync;
  * ctor takes keyword-only ``channels`` / ``num_transformer_blocks`` / ``device`` (defaults = reference values).
"""
from __future__ import annotations

import itertools
import os
from pathlib import Path

import torch

from enhanced_generator import EnhancedDiscriminator, EnhancedGenerator
from mstg_hip import dp, ops
from mstg_hip.optim import FlatAdam

LOSS_KEYS = ("d_loss", "g_loss", "cycle_loss", "identity_loss", "structure_loss")
STYLE_KEY = "style_loss"  # only present when a multi-style loss is attached (build-defined extension, see style_loss.py)


class _NoScaler:
    """fp32 path: stands in for torch.cuda.amp.GradScaler (reference :46) so code touching ``.scaler`` keeps working."""

    def scale(self, loss):
        return loss

    def step(self, optimizer):
        return optimizer.step()

    def update(self):
        return None


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


class _MeanLoss:
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, a, b):
        return self.fn(a, b)


class EnhancedCycleGAN:
    def __init__(self, pretrained_path=None, *, channels=16, num_transformer_blocks=1, device=None,
                 gradient_checkpointing=False):
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("EnhancedCycleGAN (MI355X build) needs a GPU: there is no CPU path")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.G_AB = EnhancedGenerator(channels=channels, num_transformer_blocks=num_transformer_blocks).to(self.device)
        self.G_BA = EnhancedGenerator(channels=channels, num_transformer_blocks=num_transformer_blocks).to(self.device)
        self.D_A = EnhancedDiscriminator(channels=channels).to(self.device)
        self.D_B = EnhancedDiscriminator(channels=channels).to(self.device)
        if gradient_checkpointing:  # reference :24-25 turns it on for an 8 GB-class GPU; results are identical
            self.G_AB.gradient_checkpointing_enable()
            self.G_BA.gradient_checkpointing_enable()
        if pretrained_path and os.path.exists(pretrained_path):
            checkpoint = torch.load(pretrained_path, map_location=self.device, weights_only=True)
            self.G_AB.load_state_dict(checkpoint["model_state_dict"], strict=False)
            self.G_BA.load_state_dict(checkpoint["model_state_dict"], strict=False)
        self._build_optimizers()
        self.scaler = _NoScaler()
        self.criterion_gan = _MeanLoss(ops.mse_loss)
        self.criterion_cycle = _MeanLoss(ops.l1_loss)
        self.criterion_identity = _MeanLoss(ops.l1_loss)
        self.criterion_structure = _MeanLoss(ops.l1_loss)
        self.lambda_cycle = 10.0
        self.lambda_identity = 2.0
        self.lambda_structure = 0.5
        self.batch_generator_passes = True  # see train_step_async
        self.two_streams = os.environ.get("MSTG_STREAMS", "1") != "0"  # see _train_step_async
        self.style_loss = None              # optional build-defined multi-style Gram loss on G_BA's output
        self.lambda_style = 0.0

    def attach_style_loss(self, style_refs, style_weights=(0.5, 0.3, 0.2), lambda_style=1.0, width_div=1, seed=1234):
        """BUILD-DEFINED extension (the reference has no style loss): add
        ``lambda_style * MultiStyleGramLoss(fake_A)`` to the generator objective, fake_A = G_BA(real_B) being the stylised
        image.  ``style_refs``: sequence of (N,3,H,W) tensors in [-1,1]; targets are computed once, here."""
        import style_loss as sl
        feats = sl.VGGFeatures(width_div=width_div, seed=seed).to(self.device)
        self.style_loss = sl.MultiStyleGramLoss(feats, [r.to(self.device) for r in style_refs], style_weights)
        self.lambda_style = float(lambda_style)

    def _build_optimizers(self):
        self.g_optimizer = FlatAdam(itertools.chain(self.G_AB.parameters(), self.G_BA.parameters()), lr=5e-5, betas=(0.5, 0.999))
        self.d_optimizer = FlatAdam(itertools.chain(self.D_A.parameters(), self.D_B.parameters()), lr=2e-4, betas=(0.5, 0.999))
        self._d_params = list(itertools.chain(self.D_A.parameters(), self.D_B.parameters()))

    def sync_replicas(self):
        """Make every rank start from rank 0's parameters and spectral-norm vectors (data-parallel runs)."""
        dp.broadcast_(self.g_optimizer.flat)
        dp.broadcast_(self.d_optimizer.flat)
        for D in (self.D_A, self.D_B):
            for b in D.buffers():
                dp.broadcast_(b)

    def train_step_async(self, real_A, real_B):
        # parameter gradients go straight into the two flat gradient buffers (ops.direct_param_grads), not through autograd's
        # accumulation kernels
        with ops.direct_param_grads():
            return self._train_step_async(real_A, real_B)

    # ---- two-stream execution ---------------------------------------------------------------------------------------------
    # Everything that runs through G_AB's and D_A's weights is enqueued on one side stream, everything through G_BA's and
    # D_B's on another: the two halves of a CycleGAN step are independent except where one generator's output feeds the other
    # network, and a step is ~1400 mostly small launches whose drain / fill gaps and tails then overlap (measured: both
    # generators' forward+backward 35.9 -> 31.6 ms; whole step 64.9 -> 54.9 ms).  Autograd runs every backward node on the
    # stream of its forward, so the assignment holds for the backward too; each parameter's gradient is only ever touched
    # from one stream.
    def _update_stream(self):
        if getattr(self, "_upd_stream", None) is None:
            self._upd_stream = torch.cuda.Stream(device=self.device)
        return self._upd_stream

    def _side_streams(self):
        if getattr(self, "_streams", None) is None:
            self._streams = tuple(torch.cuda.Stream(device=self.device) for _ in range(4))
        return self._streams

    def _train_step_async(self, real_A, real_B):
        G_AB, G_BA, D_A, D_B = self.G_AB, self.G_BA, self.D_A, self.D_B
        nb = real_A.shape[0]
        main = torch.cuda.current_stream(self.device)
        two = self.two_streams and not ops.KernelTimer.enabled  # per-launch timing wants one stream
        # sA: G_AB and D_A, sB: G_BA and D_B.  MSTG_STREAMS=4 gives the discriminators streams of their own (sC, sD); measured
        # slower on MI355X (1116 vs 1154 images/s): four queues contend more than the extra overlap returns.
        sA, sB, sC, sD = self._side_streams() if two else (main, main, main, main)
        if os.environ.get("MSTG_STREAMS", "1") != "4":
            sC, sD = sA, sB
        side = (sA, sB, sC, sD)

        def fork():
            if two:
                for st in side:
                    st.wait_stream(main)

        def join(*tensors):
            """main waits for the side streams; `tensors` were allocated there and are used elsewhere from now on"""
            if two:
                for st in side:
                    main.wait_stream(st)
                for t in tensors:
                    for st in (main,) + side:
                        t.record_stream(st)

        def on(st):
            return torch.cuda.stream(st) if two else _NullCtx()

        def join_backward():
            """The weight-gradient kernels of a backward run on the side streams of their forwards and write the optimizer's
            flat gradient buffer directly (ops.direct_param_grads); the all-reduce and the Adam step read that buffer on the
            main stream.  Ordered here explicitly, not through the autograd engine's end-of-backward stream sync."""
            if two:
                for st in side:
                    main.wait_stream(st)

        if self.batch_generator_passes:
            # fake_B = G_AB(real_A) (:63) and idt_B = G_AB(real_B) (:93) use the same weights (the generator optimizer only
            # steps at the end) and every op of the generator is per-sample, so one batched pass gives both, bit for bit
            # the same function of the inputs; likewise for G_BA.  Half the launches, twice the grid per launch.
            both = torch.cat([real_A, real_B], dim=0)
            fork()
            with on(sA):
                out_AB = G_AB(both)
            with on(sB):
                out_BA = G_BA(both)
            join(out_AB, out_BA, both)
            fake_B, idt_B = out_AB[:nb], out_AB[nb:]
            idt_A, fake_A = out_BA[:nb], out_BA[nb:]
        else:
            fork()
            with on(sA):
                fake_B = G_AB(real_A)
            with on(sB):
                fake_A = G_BA(real_B)
            join(fake_A, fake_B)
        # ---- discriminator update (reference :67-85)
        self.d_optimizer.zero_grad(set_to_none=True)
        fork()
        with on(sC):
            real_A_score, _ = D_A(real_A, outputs="score")  # the structure head's output is discarded here (:67-85)
            fake_A_score, _ = D_A(fake_A.detach(), outputs="score")
            dA_real, dA_fake = ops.mse_to_const(real_A_score, 1.0), ops.mse_to_const(fake_A_score, 0.0)
        with on(sD):
            real_B_score, _ = D_B(real_B, outputs="score")
            fake_B_score, _ = D_B(fake_B.detach(), outputs="score")
            dB_real, dB_fake = ops.mse_to_const(real_B_score, 1.0), ops.mse_to_const(fake_B_score, 0.0)
        join(dA_real, dA_fake, dB_real, dB_fake)
        # (dA_real + dB_real) * 0.5 + (dA_fake + dB_fake) * 0.5 (:72-81) as one weighted sum (one launch each way)
        d_loss = ops.weighted_sum((dA_real, dB_real, dA_fake, dB_fake), (0.5, 0.5, 0.5, 0.5))
        d_loss.backward()
        # The discriminator exchange + update (one all-reduce over the flat gradient buffer, one fused Adam launch) runs on a
        # stream of its own: in the generator phase only the discriminator forwards depend on it, so the two cycle reconstructions
        # (the largest launches of the step) start at once instead of waiting out the collective's latency on xGMI.
        upd = self._update_stream() if two else main
        if two:
            upd.wait_stream(main)
            for st in side:
                upd.wait_stream(st)  # the weight-gradient kernels of this backward ran on the side streams of their forwards
        with on(upd):
            dp.allreduce_mean_(self.d_optimizer.grad)
            self.d_optimizer.step()
        # ---- generator update (reference :88-123); D weights take no gradient here (it would be discarded)
        self.g_optimizer.zero_grad(set_to_none=True)
        for p in self._d_params:
            p.requires_grad_(False)
        try:
            if not self.batch_generator_passes:
                fork()
                with on(sB):
                    idt_A = G_BA(real_A)
                with on(sA):
                    idt_B = G_AB(real_B)
                join(idt_A, idt_B)
            idA_l, idB_l = ops.l1_loss(idt_A, real_A), ops.l1_loss(idt_B, real_B)
            # The reference runs D on the fakes a second time for the structure heads (:110-113).  Outputs are the
            # same function of the same inputs except for the spectral-norm power iteration that every train-mode
            # forward performs, so the call count per D is kept at the reference's 5 per step, in the reference's order.
            fork()
            with on(sA):
                recon_B = G_AB(fake_A)
                cB = ops.l1_loss(recon_B, real_B)
            # enqueue order on one GPU: stream A gets its large reconstruction first and its discriminator's small launches second,
            # stream B the other way round, so that small work overlaps large work rather than small with small.  With more than
            # one rank both streams start with their reconstruction: the discriminator passes wait for the update stream (the
            # all-reduce), the reconstructions do not.
            recon_first = dp.world_size() > 1
            if recon_first:
                with on(sB):
                    recon_A = G_BA(fake_B)
                    cA = ops.l1_loss(recon_A, real_A)
            if two:
                sC.wait_stream(upd)
                sD.wait_stream(upd)
            with on(sD):
                fake_B_score, _ = D_B(fake_B, outputs="score")  # its structure output is overwritten two lines down (:110-113)
                gB = ops.mse_to_const(fake_B_score, 1.0)
                with torch.no_grad():
                    _, real_B_struct = D_B(real_B, outputs="struct")
                _, fake_B_struct = D_B(fake_B, outputs="struct")
                sB_l = ops.l1_loss(real_B_struct, fake_B_struct)
            if not recon_first:
                with on(sB):
                    recon_A = G_BA(fake_B)
                    cA = ops.l1_loss(recon_A, real_A)
            with on(sC):
                fake_A_score, _ = D_A(fake_A, outputs="score")
                gA = ops.mse_to_const(fake_A_score, 1.0)
                with torch.no_grad():
                    _, real_A_struct = D_A(real_A, outputs="struct")
                _, fake_A_struct = D_A(fake_A, outputs="struct")
                sA_l = ops.l1_loss(real_A_struct, fake_A_struct)
            join(gA, gB, cA, cB, sA_l, sB_l)
            # g + cycle * lambda + identity * lambda + structure * lambda (:95-118): the total as ONE weighted sum over the eight terms
            # (one launch forward, one backward); the four reported components come out of the same launch, without a graph
            lc, li, ls = self.lambda_cycle, self.lambda_identity, self.lambda_structure
            total_g_loss, parts = ops.weighted_sum((gA, gB, cA, cB, idA_l, idB_l, sA_l, sB_l), (1.0, 1.0, lc, lc, li, li, ls, ls),
                                                   report=((1.0, 1.0, 0, 0, 0, 0, 0, 0), (0, 0, lc, lc, 0, 0, 0, 0), (0, 0, 0, 0, li, li, 0, 0),
                                                           (0, 0, 0, 0, 0, 0, ls, ls)))
            g_loss, cycle_loss, identity_loss, structure_loss = parts[0], parts[1], parts[2], parts[3]
            style = None
            if self.style_loss is not None:
                style = self.style_loss(fake_A) * self.lambda_style
                total_g_loss = total_g_loss + style
            total_g_loss.backward()
        finally:
            for p in self._d_params:
                p.requires_grad_(True)
        join_backward()
        dp.allreduce_mean_(self.g_optimizer.grad)
        self.g_optimizer.step()
        out = [d_loss.detach(), g_loss.detach(), cycle_loss.detach(), identity_loss.detach(), structure_loss.detach()]
        if style is not None:
            out.append(style.detach())
        return torch.stack(out)

    def train_step(self, real_A, real_B):
        vals = self.train_step_async(real_A, real_B).tolist()  # one device sync
        return dict(zip(LOSS_KEYS + ((STYLE_KEY,) if self.style_loss is not None else ()), vals))

    def save_models(self, save_dir, epoch):  # reference :133-152
        save_path = Path(save_dir)
        save_path.mkdir(parents=True, exist_ok=True)

        def sd(m):  # parameters are views of an optimizer's flat buffer: torch.save would serialise the WHOLE buffer per view
            return {k: v.detach().clone() for k, v in m.state_dict().items()}

        torch.save({"epoch": epoch, "G_AB_state_dict": sd(self.G_AB)}, save_path / f"G_AB_epoch_{epoch}.pth")
        torch.save({"epoch": epoch, "G_BA_state_dict": sd(self.G_BA)}, save_path / f"G_BA_epoch_{epoch}.pth")
        torch.save({"epoch": epoch, "D_A_state_dict": sd(self.D_A), "D_B_state_dict": sd(self.D_B)},
                   save_path / f"discriminators_epoch_{epoch}.pth")

    def load_models(self, save_dir, epoch):
        """Inverse of save_models (the reference has no resume path for the GAN trainer; its inference
        scripts read the same files: advanced_transform.py:14-32).  Strict; values are copied INTO the flat-buffer views."""
        save_path = Path(save_dir)
        for fname, pairs in ((f"G_AB_epoch_{epoch}.pth", (("G_AB_state_dict", self.G_AB),)),
                             (f"G_BA_epoch_{epoch}.pth", (("G_BA_state_dict", self.G_BA),)),
                             (f"discriminators_epoch_{epoch}.pth", (("D_A_state_dict", self.D_A), ("D_B_state_dict", self.D_B)))):
            ckpt = torch.load(save_path / fname, map_location=self.device, weights_only=True)
            for key, module in pairs:
                # a file convert_model.py has flattened (convert_model.py:12-29) holds the bare state dict of its one network
                sd = ckpt[key] if isinstance(ckpt, dict) and key in ckpt else (ckpt if len(pairs) == 1 else None)
                if sd is None:
                    raise KeyError(f"{fname}: no '{key}' entry")
                module.load_state_dict(sd, strict=True)
        return epoch


def train(data_root, save_dir, pretrained_path=None, num_epochs=200, batch_size=1, *, channels=16, num_transformer_blocks=1,
          datasets=None, log_every=10, save_every=20):
    """The reference's training loop (enhanced_train.py:154-208) on the MI355X step: two ``MonetPhotoDataset`` (transform on the
    GPU, pretrain.py drop-in), zipped loaders, ``train_step`` per pair, ``save_models`` every 20 epochs.

    Kept from the reference on purpose: the generators see ``batch[0]`` of the dataset triple, i.e. the MASKED image
    (:184-185 with pretrain.py:56-57).  Different on purpose: ``batch_size`` is honoured (the reference ignores its argument and
    always uses 1, :167-170 -- the default reproduces that); errors propagate instead of ending training silently (:204-208);
    losses are read back (one sync) only when they are printed."""
    from pretrain import DeviceLoader, MonetPhotoDataset, set_seed  # the drop-in module next to this file
    set_seed(42)
    model = EnhancedCycleGAN(pretrained_path, channels=channels, num_transformer_blocks=num_transformer_blocks)
    if datasets is None:
        datasets = (MonetPhotoDataset(data_root, domain="A", device=model.device), MonetPhotoDataset(data_root, domain="B", device=model.device))
    monet_loader = DeviceLoader(datasets[0], batch_size=batch_size, shuffle=True, drop_last=True)
    photo_loader = DeviceLoader(datasets[1], batch_size=batch_size, shuffle=True, drop_last=True)
    history = []
    for epoch in range(num_epochs):
        for i, (monet_batch, photo_batch) in enumerate(zip(monet_loader, photo_loader)):
            real_A, real_B = monet_batch[0], photo_batch[0]
            losses = model.train_step_async(real_A, real_B)
            if (i + 1) % log_every == 0:
                vals = dict(zip(LOSS_KEYS, losses.tolist()))
                history.append((epoch, i, vals))
                print(f"Epoch [{epoch + 1}/{num_epochs}], Step [{i + 1}/{len(monet_loader)}] "
                      + " ".join(f"{k}: {v:.4f}" for k, v in vals.items()))
        if (epoch + 1) % save_every == 0:
            model.save_models(save_dir, epoch + 1)
    return model, history
