"""Device-side image pipeline around the generator: what PIL / torchvision / numpy do on the CPU in the
reference's dataset (pretrain.py:20-57) and in ``process_cyclegan`` (batch_process_images.py:176-245), on uint8 HWC tensors that
live on the GPU.  Resampling is Pillow's algorithm bit for bit (csrc/image.hip); decoding / encoding image FILES stays with PIL on
the host -- this module starts from and ends with uint8 arrays.
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache

import numpy as np
import torch

from . import _lib
from .ops import _p, _stream

BILINEAR, LANCZOS = 0, 1


@lru_cache(maxsize=256)
def _coeff_tables_host(in_size: int, out_size: int, filt: int):
    lib = _lib.load()
    ks = lib.mstg_resample_ksize(in_size, out_size, filt)
    if ks <= 0:
        raise RuntimeError(f"mstg_hip resample: bad sizes {in_size} -> {out_size}")
    kk = np.zeros((out_size, ks), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    _lib.check(lib.mstg_resample_coeffs(in_size, out_size, filt, kk.ctypes.data, bounds.ctypes.data), "mstg_resample_coeffs")
    return ks, kk, bounds


_dev_tables = {}


def _coeff_tables(in_size, out_size, filt, device):
    key = (in_size, out_size, filt, str(device))
    if key not in _dev_tables:
        ks, kk, bounds = _coeff_tables_host(in_size, out_size, filt)
        _dev_tables[key] = (ks, torch.from_numpy(kk).to(device), torch.from_numpy(bounds).to(device), bounds)
    return _dev_tables[key]


def _req_u8(img: torch.Tensor) -> torch.Tensor:
    if not img.is_cuda or img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
        raise RuntimeError("mstg_hip image: expected a uint8 (H, W, 3) tensor on the GPU")
    return img.contiguous()


def resize_u8(img: torch.Tensor, size, filt: int = BILINEAR) -> torch.Tensor:
    """``PIL.Image.resize((w, h), filter)`` of an (H, W, 3) uint8 image; ``size`` = (new_w, new_h) like PIL."""
    img = _req_u8(img)
    H, W = img.shape[:2]
    new_w, new_h = int(size[0]), int(size[1])
    lib = _lib.load()
    need_h, need_v = new_w != W, new_h != H
    if not need_h and not need_v:
        return img.clone()
    cur, y_first = img, 0
    if need_v:
        ksv, kkv, bv_dev, bv_host = _coeff_tables(H, new_h, filt, img.device)
        y_first = int(bv_host[0, 0])
        y_last = int(bv_host[new_h - 1, 0] + bv_host[new_h - 1, 1])
    else:
        y_last = H
    if need_h:
        ksh, kkh, bh_dev, _ = _coeff_tables(W, new_w, filt, img.device)
        rows = y_last - y_first if need_v else H
        tmp = torch.empty((rows, new_w, 3), dtype=torch.uint8, device=img.device)
        _lib.check(lib.mstg_resample_h_u8(_p(cur), _p(tmp), W, y_first if need_v else 0, rows, new_w, ksh, _p(kkh), _p(bh_dev), _stream()),
                   "mstg_resample_h_u8")
        cur = tmp
    if need_v:
        bounds = bv_dev
        if need_h and y_first:  # Pillow shifts the vertical bounds by the first row the horizontal pass kept
            bounds = bv_dev.clone()
            bounds[:, 0] -= y_first
        out = torch.empty((new_h, cur.shape[1], 3), dtype=torch.uint8, device=img.device)
        _lib.check(lib.mstg_resample_v_u8(_p(cur), _p(out), cur.shape[1], new_h, ksv, _p(kkv), _p(bounds), _stream()), "mstg_resample_v_u8")
        cur = out
    return cur


def paste_u8(src, window, dst_hw, at, fill=255) -> torch.Tensor:
    """New (dh, dw) canvas filled with ``fill`` with ``src[window]`` pasted at ``at`` = (y, x); window = (y0, x0, h, w)."""
    src = _req_u8(src)
    y0, x0, h, w = window
    dst = torch.empty((dst_hw[0], dst_hw[1], 3), dtype=torch.uint8, device=src.device)
    _lib.check(_lib.load().mstg_paste_u8(_p(src), src.shape[0], src.shape[1], y0, x0, h, w, _p(dst), dst_hw[0], dst_hw[1], at[0], at[1],
                                         fill, _stream()), "mstg_paste_u8")
    return dst


def crop_u8(src, left, top, right, bottom) -> torch.Tensor:
    """``Image.crop((left, top, right, bottom))`` for a box inside the image."""
    return paste_u8(src, (top, left, bottom - top, right - left), (bottom - top, right - left), (0, 0), fill=0)


def to_tensor(img, window=None, grid_mask=None):
    """ToTensor + Normalize(0.5, 0.5) -> (3, H, W) fp32 in [-1, 1].  With ``grid_mask`` (64-bit int, bit i*8+j = keep cell (i, j))
    returns (masked_image, image, mask) like MonetPhotoDataset.__getitem__ (pretrain.py:56-57)."""
    img = _req_u8(img)
    y0, x0, H, W = window if window is not None else (0, 0, img.shape[0], img.shape[1])
    out = torch.empty((3, H, W), dtype=torch.float32, device=img.device)
    lib = _lib.load()
    if grid_mask is None:
        _lib.check(lib.mstg_u8_to_tensor(_p(img), img.shape[0], img.shape[1], y0, x0, H, W, _p(out), None, None, 0, 0, _stream()),
                   "mstg_u8_to_tensor")
        return out
    image, mask = torch.empty_like(out), torch.empty_like(out)
    _lib.check(lib.mstg_u8_to_tensor(_p(img), img.shape[0], img.shape[1], y0, x0, H, W, _p(out), _p(image), _p(mask),
                                     int(grid_mask) & (2 ** 64 - 1), 1, _stream()), "mstg_u8_to_tensor")
    return out, image, mask


def to_u8(y: torch.Tensor) -> torch.Tensor:
    """(3, H, W) generator output -> (H, W, 3) uint8: (y + 1) / 2, clamp, * 255, astype(uint8) (batch_process_images.py:213-217)."""
    if not y.is_cuda or y.dim() != 3 or y.shape[0] != 3:
        raise RuntimeError("mstg_hip image: expected a (3, H, W) tensor on the GPU")
    y = y.float().contiguous()
    out = torch.empty((y.shape[1], y.shape[2], 3), dtype=torch.uint8, device=y.device)
    _lib.check(_lib.load().mstg_tensor_to_u8(_p(y), y.shape[1], y.shape[2], _p(out), _stream()), "mstg_tensor_to_u8")
    return out


def blend_u8(orig: torch.Tensor, styled: torch.Tensor, strength=None, weight_map=None) -> torch.Tensor:
    """``np.clip(orig * (1 - w) + styled * w, 0, 255).astype(np.uint8)`` on two (H, W, 3) uint8 images, bit for bit as numpy
    evaluates it in float64: w = the scalar ``strength`` (process_local_style mode 'simple', batch_process_images.py:304-312) or a
    float64 (H, W) ``weight_map`` (the 'enhanced' / 'advanced' blend of :340-342, :386-387 given a mask made on the host)."""
    orig, styled = _req_u8(orig), _req_u8(styled)
    if orig.shape != styled.shape:
        raise RuntimeError(f"mstg_hip blend: shapes differ {tuple(orig.shape)} vs {tuple(styled.shape)}")
    if (strength is None) == (weight_map is None):
        raise RuntimeError("mstg_hip blend: give either strength or weight_map")
    H, W = orig.shape[:2]
    out = torch.empty_like(orig)
    if weight_map is not None:
        wm = torch.as_tensor(weight_map)
        if tuple(wm.shape) != (H, W):
            raise RuntimeError(f"mstg_hip blend: weight map {tuple(wm.shape)} does not match the image {H}x{W}")
        wm = wm.to(device=orig.device, dtype=torch.float64).contiguous()
        w0 = w1 = 0.0
    else:
        wm, w1 = None, float(strength)
        w0 = 1 - w1  # Python's double subtraction, as in the reference's expression
    _lib.check(_lib.load().mstg_blend_u8(_p(orig), _p(styled), w0, w1, _p(wm), _p(out), H, W, _stream()), "mstg_blend_u8")
    return out


# ---- the callers, restated on the device ----------------------------------------------------------------------------------
def dataset_item(img_u8: torch.Tensor, grid_mask: int, img_size: int = 256):
    """MonetPhotoDataset.__getitem__ (pretrain.py:41-57) from a decoded uint8 image on the GPU: Resize(img_size) (shorter side,
    bilinear) -> CenterCrop -> ToTensor -> Normalize -> 8x8-grid mask.  Returns (masked_image, image, mask)."""
    H, W = img_u8.shape[:2]
    if W <= H:  # torchvision Resize(int): the smaller edge becomes img_size, the other int(size * long / short)
        new_w, new_h = img_size, int(img_size * H / W)
    else:
        new_h, new_w = img_size, int(img_size * W / H)
    r = resize_u8(img_u8, (new_w, new_h), BILINEAR) if (new_w, new_h) != (W, H) else img_u8
    top, left = int(round((new_h - img_size) / 2.0)), int(round((new_w - img_size) / 2.0))  # torchvision center_crop
    return to_tensor(r, (top, left, img_size, img_size), grid_mask)


def letterbox(img_u8: torch.Tensor, target=256):
    """batch_process_images.py:183-199: aspect-preserving LANCZOS resize onto a white target x target canvas -> (canvas, geometry)."""
    height, width = img_u8.shape[:2]
    if width > height:
        new_width, new_height = target, int(height * (target / width))
    else:
        new_height, new_width = target, int(width * (target / height))
    resized = resize_u8(img_u8, (new_width, new_height), LANCZOS)
    off_x, off_y = (target - new_width) // 2, (target - new_height) // 2
    canvas = paste_u8(resized, (0, 0, new_height, new_width), (target, target), (off_y, off_x), fill=255)
    return canvas, (width, height)


def process_cyclegan(model, img_u8: torch.Tensor, target=256) -> torch.Tensor:
    """``process_cyclegan`` of the reference (batch_process_images.py:176-236) without the file I/O: decoded uint8 image in, uint8
    image out, everything in between on the GPU (letterbox, forward, output conversion, crop back, resize back)."""
    canvas, (width, height) = letterbox(img_u8, target)
    x = to_tensor(canvas).unsqueeze(0)
    with torch.no_grad():
        y = model(x)
    out = to_u8(y[0])
    if width != height:
        aspect = width / height
        if aspect > 1:
            crop_w, crop_h = target, int(target / aspect)
        else:
            crop_h, crop_w = target, int(target * aspect)
        left, top = (target - crop_w) // 2, (target - crop_h) // 2
        out = crop_u8(out, left, top, left + crop_w, top + crop_h)
    if width * height <= 1024 * 1024:
        out = resize_u8(out, (width, height), LANCZOS)
    return out


def process_local_style(model, img_u8: torch.Tensor, mode="simple", strength=0.8, weight_map=None, target=256) -> torch.Tensor:
    """``process_local_style`` of the reference (batch_process_images.py:255-441) without the file I/O, on the GPU end to end for
    the modes that are byte arithmetic: 'simple' (strength blend of the letterboxed original with the styled image, :304-312),
    'weight_map' (the per-pixel blend of the 'enhanced' mode, :340-342 + :352, with the (target, target) float64 weight map the
    caller built -- cv2.Canny / scipy.gaussian_filter / the sky detector are host code outside this library), and any other mode
    name = the reference's default branch (the styled image itself, :399-401).  Crop back to the aspect ratio and LANCZOS resize
    back as :403-429."""
    canvas, (width, height) = letterbox(img_u8, target)
    x = to_tensor(canvas).unsqueeze(0)
    with torch.no_grad():
        y = model(x)
    styled = to_u8(y[0])
    if mode == "simple":
        out = blend_u8(canvas, styled, strength=strength)
    elif mode == "weight_map":
        out = blend_u8(canvas, styled, weight_map=weight_map)
    else:
        out = styled
    aspect = width / height
    if aspect != 1.0:
        if aspect > 1:
            crop_w, crop_h = target, int(target / aspect)
        else:
            crop_h, crop_w = target, int(target * aspect)
        crop_w, crop_h = min(crop_w, target), min(crop_h, target)
        left, top = (target - crop_w) // 2, (target - crop_h) // 2
        out = crop_u8(out, left, top, left + crop_w, top + crop_h)
    if width * height <= 1024 * 1024:
        out = resize_u8(out, (width, height), LANCZOS)
    return out
