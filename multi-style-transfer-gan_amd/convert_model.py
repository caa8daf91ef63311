"""Drop-in for the reference's ``convert_model.py``: flatten a checkpoint wrapper to a bare state dict.

``flatten_checkpoint`` restates convert_model.py:12-29 (which key wins, in the reference's order); ``convert_model`` is its
file-to-file form (:5-37); ``load_generator`` is the loading convention every inference caller of the reference repeats
(direct_transform.py:10-42, advanced_transform.py:12-36, batch_process_images.py:60-124): accept a wrapper or a bare state dict,
infer ``channels`` from ``initial.0.weight`` and build ``EnhancedGenerator(channels, num_transformer_blocks=1)``.

Files are read with ``weights_only=True``: nothing stored in a checkpoint is executed.
"""
from __future__ import annotations

import torch


def flatten_checkpoint(checkpoint):
    """convert_model.py:12-29: G_AB_state_dict, else G_BA_state_dict, else (for a dict with 'epoch') state_dict / model_state_dict /
    every entry that is neither 'epoch' nor a 'G_*' key, else the object itself."""
    if isinstance(checkpoint, dict) and "G_AB_state_dict" in checkpoint:
        return checkpoint["G_AB_state_dict"]
    if isinstance(checkpoint, dict) and "G_BA_state_dict" in checkpoint:
        return checkpoint["G_BA_state_dict"]
    if isinstance(checkpoint, dict) and "epoch" in checkpoint:
        if "state_dict" in checkpoint:
            return checkpoint["state_dict"]
        if "model_state_dict" in checkpoint:
            return checkpoint["model_state_dict"]
        return {k: v for k, v in checkpoint.items() if k != "epoch" and not k.startswith("G_")}
    return checkpoint


def convert_model(input_path, output_path) -> bool:
    """convert_model.py:5-37 (prints and returns False on failure, like the reference)."""
    try:
        checkpoint = torch.load(input_path, map_location="cpu", weights_only=True)
        torch.save(flatten_checkpoint(checkpoint), output_path)
        print(f"model converted and saved to {output_path}")
        return True
    except Exception as e:  # the reference swallows every error here (:35-37)
        print(f"conversion failed: {e}")
        return False


def load_generator(model_path, device=None, num_transformer_blocks: int = 1, direction: str = "AB"):
    """direct_transform.py:10-42: wrapper or bare state dict -> EnhancedGenerator on ``device`` in eval mode.  ``direction`` picks
    G_AB / G_BA when the file holds the other key too (batch_process_images.py:100-117); channels come from initial.0.weight."""
    from enhanced_generator import EnhancedGenerator
    checkpoint = torch.load(model_path, map_location="cpu", weights_only=True)
    key = f"G_{direction}_state_dict"
    state_dict = checkpoint[key] if isinstance(checkpoint, dict) and key in checkpoint else flatten_checkpoint(checkpoint)
    channels = int(state_dict["initial.0.weight"].shape[0]) if "initial.0.weight" in state_dict else 16  # :25-30, default :12
    model = EnhancedGenerator(channels=channels, num_transformer_blocks=num_transformer_blocks)
    model.load_state_dict(state_dict)
    if device is not None:
        model = model.to(device)
    model.eval()
    return model


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="flatten a checkpoint wrapper to a bare state dict")
    ap.add_argument("--input", type=str, required=True)
    ap.add_argument("--output", type=str, required=True)
    a = ap.parse_args()
    convert_model(a.input, a.output)
