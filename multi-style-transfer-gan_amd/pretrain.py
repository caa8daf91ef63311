"""Drop-in for the reference's ``pretrain`` module (pretrain.py) on the MI355X kernels: ``set_seed``, ``MonetPhotoDataset``,
``Generator`` and ``train`` (masked-image L1 pre-training of the plain CycleGAN generator).

Same names and call shapes as the reference (``enhanced_train.py:11`` does ``from pretrain import MonetPhotoDataset, set_seed``).
What differs, deliberately:
  * ``MonetPhotoDataset`` decodes the image FILE with PIL on the host (that is I/O) and then does everything else --
    Resize(img_size) / CenterCrop / ToTensor / Normalize / the 8x8-grid 40 % mask (pretrain.py:32-57) -- on the GPU, bit-exact
    against Pillow's resampling (mstg_hip/image.py).  Items are CUDA tensors; the 64 ``random.random()`` draws per item come
    from Python's ``random`` in the reference's order, so a seeded run masks the same cells.
  * ``train`` keeps the reference's loop (two domains per epoch, Adam 2e-4 / (0.5, 0.999), CosineAnnealingLR(T_max=num_epochs,
    eta_min=1e-6), clip_grad_norm_(1.0), checkpoint every 50 epochs with the reference's keys) but steps through
    ``PretrainStep``: HIP forward / backward, masked L1 loss, gradient clipping and Adam on one flat buffer.  Arithmetic is fp32
    (the reference's ``autocast`` is a no-op on its CPU path, the parity target); errors propagate instead of being swallowed
    by the reference's blanket ``except`` (pretrain.py:221-225).
"""
from __future__ import annotations

import math
import os
import random
from pathlib import Path

import numpy as np
import torch

from mstg_hip import image as dimg
from mstg_hip import ops
from mstg_hip.optim import FlatAdam
from plain_generator import Generator  # noqa: F401  (re-exported under the reference's name, pretrain.py:60-97)


def set_seed(seed=42):  # pretrain.py:13-17
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def draw_grid_mask() -> int:
    """The 64 draws of pretrain.py:47-50 (row-major, ``random.random() < 0.4`` masks the cell) as a keep-bitmask."""
    grid = 0
    for i in range(8):
        for j in range(8):
            if not random.random() < 0.4:
                grid |= 1 << (i * 8 + j)
    return grid


class MonetPhotoDataset:
    """``MonetPhotoDataset(root_dir, domain, split='train', img_size=256)`` (pretrain.py:20-57) with the transform on the GPU.
    ``arrays`` (list of (H, W, 3) uint8 numpy arrays) replaces the directory listing for synthetic data / tests."""

    def __init__(self, root_dir=None, domain="A", split="train", img_size=256, device=None, arrays=None):
        self.root_dir = None if root_dir is None else Path(root_dir)
        self.domain, self.split, self.img_size = domain, split, img_size
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("MonetPhotoDataset (MI355X build) needs a GPU: the transform has no CPU path")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.arrays = arrays
        if arrays is None:
            if self.root_dir is None:
                raise ValueError("MonetPhotoDataset: give root_dir or arrays")
            self.image_paths = list((self.root_dir / f"{split}{domain}").glob("*.jpg"))
            self.image_paths.extend(list((self.root_dir / f"{split}{domain}").glob("*.png")))
        else:
            self.image_paths = [None] * len(arrays)

    def __len__(self):
        return len(self.image_paths)

    def _decode(self, idx) -> np.ndarray:
        if self.arrays is not None:
            return np.ascontiguousarray(self.arrays[idx], dtype=np.uint8)
        from PIL import Image  # file decoding is host I/O; everything after it is on the GPU
        return np.asarray(Image.open(self.image_paths[idx]).convert("RGB"))

    def __getitem__(self, idx):
        img = torch.from_numpy(self._decode(idx)).to(self.device)
        grid = draw_grid_mask()
        return dimg.dataset_item(img, grid, self.img_size)  # (masked_image, image, mask)


class DeviceLoader:
    """What ``DataLoader(dataset, batch_size, shuffle=True, drop_last=...)`` is to the reference's loops, for items that already
    live on the GPU: shuffles with Python's ``random``, stacks ``batch_size`` items per field."""

    def __init__(self, dataset, batch_size=1, shuffle=True, drop_last=False):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, batch_size, shuffle, drop_last

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = list(range(len(self.dataset)))
        if self.shuffle:
            random.shuffle(order)
        for b in range(len(self)):
            items = [self.dataset[i] for i in order[b * self.batch_size:(b + 1) * self.batch_size]]
            yield tuple(torch.stack(f) for f in zip(*items))


class CosineLR:
    """torch.optim.lr_scheduler.CosineAnnealingLR(T_max, eta_min) in closed form, for FlatAdam's single parameter group."""

    def __init__(self, optimizer, T_max, eta_min=0.0):
        self.optimizer, self.T_max, self.eta_min = optimizer, T_max, eta_min
        self.base_lr = optimizer.param_groups[0]["lr"]
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        lr = self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * self.last_epoch / self.T_max)) / 2
        self.optimizer.param_groups[0]["lr"] = lr

    def get_last_lr(self):
        return [self.optimizer.param_groups[0]["lr"]]

    def state_dict(self):
        """The keys torch.optim.lr_scheduler.CosineAnnealingLR.state_dict() holds (it is ``__dict__`` minus the optimizer), so that
        pretrain_resume.py:149-150 can load it into the torch scheduler."""
        return {"T_max": self.T_max, "eta_min": self.eta_min, "base_lrs": [self.base_lr], "last_epoch": self.last_epoch,
                "_step_count": self.last_epoch + 1, "_get_lr_called_within_step": False, "_last_lr": self.get_last_lr()}

    def load_state_dict(self, sd):
        """torch's CosineAnnealingLR state (``base_lrs``) or the round-2 layout of this class (``base_lr``)."""
        self.T_max, self.eta_min, self.last_epoch = sd["T_max"], sd["eta_min"], int(sd["last_epoch"])
        self.base_lr = sd["base_lrs"][0] if "base_lrs" in sd else sd["base_lr"]
        if self.last_epoch:
            self.last_epoch -= 1
            self.step()
        else:
            self.optimizer.param_groups[0]["lr"] = self.base_lr


class PretrainStep:
    """One optimisation step of pretrain.py:154-166: zero_grad, forward, L1(gen * (1 - mask), real * (1 - mask)), backward,
    clip_grad_norm_(1.0), Adam step.  Returns the loss as a 0-dim device tensor (no host sync)."""

    def __init__(self, generator, lr=2e-4, max_norm=1.0):
        self.generator = generator
        self.optimizer = FlatAdam(generator.parameters(), lr=lr, betas=(0.5, 0.999))
        self.max_norm = max_norm
        self.last_grad_norm = None

    def __call__(self, masked_imgs, real_imgs, masks):
        self.optimizer.zero_grad()
        with ops.direct_param_grads():
            generated = self.generator(masked_imgs)
            loss = ops.masked_l1_loss(generated, real_imgs, masks)
            loss.backward()
        self.last_grad_norm = ops.clip_grad_norm_flat_(self.optimizer.grad, self.max_norm)
        self.optimizer.step()
        return loss.detach()


def load_checkpoint(path, generator, optimizer=None, scheduler=None, device=None):
    """pretrain_resume.py:134-157: ``model_state_dict`` (or a bare state dict), then -- when present -- optimizer and scheduler
    state, start_epoch = checkpoint['epoch'] + 1.  Reads with weights_only=True (nothing in the file is executed)."""
    ckpt = torch.load(path, map_location=device, weights_only=True)
    generator.load_state_dict(ckpt["model_state_dict"] if "model_state_dict" in ckpt else ckpt)
    start_epoch = 0
    if isinstance(ckpt, dict):
        if optimizer is not None and "optimizer_state_dict" in ckpt:
            optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        if scheduler is not None and "scheduler_state_dict" in ckpt:
            scheduler.load_state_dict(ckpt["scheduler_state_dict"])
        if "epoch" in ckpt:
            start_epoch = int(ckpt["epoch"]) + 1
    return start_epoch


def train(data_root, save_dir, num_epochs=200, batch_size=1, lr=2e-4, channels=64, datasets=None, log_every=10, resume_path=None,
          continue_epochs=False, save_every=50):
    """pretrain.py:99-230, and with ``resume_path`` pretrain_resume.py:134-157.  ``datasets`` = (monet_dataset, photo_dataset)
    overrides the directory-backed ones.  As in the reference, a resumed run restores model / optimizer / scheduler and reports
    ``start_epoch`` but its loop still counts from 0 (pretrain_resume.py:166); ``continue_epochs=True`` starts the loop at
    ``start_epoch`` instead."""
    set_seed()
    if not torch.cuda.is_available():
        raise RuntimeError("pretrain.train (MI355X build) needs a GPU: there is no CPU path")
    device = torch.device("cuda", torch.cuda.current_device())
    os.makedirs(save_dir, exist_ok=True)
    if datasets is None:
        datasets = (MonetPhotoDataset(data_root, domain="A", device=device), MonetPhotoDataset(data_root, domain="B", device=device))
    monet_loader = DeviceLoader(datasets[0], batch_size=batch_size, shuffle=True, drop_last=True)
    photo_loader = DeviceLoader(datasets[1], batch_size=batch_size, shuffle=True, drop_last=True)
    generator = Generator(channels=channels).to(device)
    step = PretrainStep(generator, lr=lr)
    scheduler = CosineLR(step.optimizer, T_max=num_epochs, eta_min=1e-6)
    start_epoch = 0
    if resume_path is not None and os.path.exists(resume_path):
        start_epoch = load_checkpoint(resume_path, generator, step.optimizer, scheduler, device)
        print(f"[INFO] resumed from {resume_path}: start_epoch {start_epoch}")
    history = []
    for epoch in range(start_epoch if continue_epochs else 0, num_epochs):
        generator.train()
        for name, loader in (("monet", monet_loader), ("photo", photo_loader)):
            running = None
            for i, (masked_imgs, real_imgs, masks) in enumerate(loader):
                loss = step(masked_imgs, real_imgs, masks)
                running = loss if running is None else running + loss
                if (i + 1) % log_every == 0:
                    print(f"  [{name}] Batch [{i + 1}/{len(loader)}], Loss: {float(running) / log_every:.4f}")
                    running = None
            history.append((epoch, name, float(loss)))
        scheduler.step()
        if (epoch + 1) % save_every == 0:
            torch.save({"epoch": epoch,
                        "model_state_dict": {k: v.detach().clone() for k, v in generator.state_dict().items()},
                        "optimizer_state_dict": step.optimizer.state_dict(),
                        "scheduler_state_dict": scheduler.state_dict(),
                        "loss": 0},  # the reference never accumulates total_loss (pretrain.py:146,215)
                       os.path.join(save_dir, f"generator_pretrain_epoch_{epoch + 1}.pth"))
        print(f"Epoch [{epoch + 1}/{num_epochs}] lr {scheduler.get_last_lr()[0]:.6f}")
    return generator, history
