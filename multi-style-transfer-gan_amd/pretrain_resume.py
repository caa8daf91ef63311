"""Drop-in for the reference's ``pretrain_resume.py``: pretrain.py with ``channels=128`` and a ``resume_path``
(pretrain_resume.py:99-157).  Everything lives in ``pretrain``; this module keeps the import name and the defaults."""
from __future__ import annotations

from pretrain import DeviceLoader, Generator, MonetPhotoDataset, load_checkpoint, set_seed  # noqa: F401
from pretrain import train as _train


def train(data_root, save_dir, num_epochs=200, batch_size=1, lr=2e-4, resume_path=None, **kw):
    """pretrain_resume.py:99: ``Generator(channels=128)`` (:127); model / optimizer / scheduler restored from ``resume_path``
    (:134-157).  Checkpoints written by the reference (torch.optim.Adam / CosineAnnealingLR state) and by this package load."""
    kw.setdefault("channels", 128)
    return _train(data_root, save_dir, num_epochs=num_epochs, batch_size=batch_size, lr=lr, resume_path=resume_path, **kw)
