"""CPU restatement of the image pre/post-processing either side of the generator -- TEST INFRASTRUCTURE ONLY.

Two layers:
  * ``pil_*``: what the reference literally calls -- ``PIL.Image.resize`` / ``Image.new`` / ``paste`` / ``crop`` -- composed as in
    MonetPhotoDataset (pretrain.py:32-57; torchvision's Resize(int) / CenterCrop / ToTensor / Normalize restated from their
    documented definitions, torchvision itself is not installed) and in process_cyclegan (batch_process_images.py:176-236).
    Pillow is the reference's own third-party dependency (un-pinned; this image has 12.2.0).
  * ``resample_numpy``: Pillow's two-pass 8-bit resampling (src/libImaging/Resample.c: precompute_coeffs,
    normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc) restated in numpy; pinned by
    tests/test_image_cpu.py against ``PIL.Image.resize`` bit for bit.  This is what the HIP kernels are compared with where
    Pillow is not importable.
"""
from __future__ import annotations

import math

import numpy as np

BILINEAR, LANCZOS = 0, 1
PRECISION_BITS = 32 - 8 - 2


def _filter(filt):
    if filt == BILINEAR:
        return 1.0, lambda x: (1.0 - abs(x)) if abs(x) < 1.0 else 0.0

    def sinc(x):
        if x == 0.0:
            return 1.0
        x = x * math.pi
        return math.sin(x) / x
    return 3.0, lambda x: sinc(x) * sinc(x / 3) if -3.0 <= x < 3.0 else 0.0


def coeffs(in_size: int, out_size: int, filt: int):
    """precompute_coeffs + normalize_coeffs_8bpc -> (ksize, kk int32 [out][ksize], bounds int32 [out][2])"""
    support0, f = _filter(filt)
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(k[i] for i in range(xmax)) if xmax else 0.0
        ww = 0.0
        for v in k:
            ww += v
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, kk, bounds


def _clip8(a):
    return np.clip(a >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_numpy(img: np.ndarray, size, filt: int) -> np.ndarray:
    """``Image.resize((w, h), filt)`` of an (H, W, 3) uint8 array, integer arithmetic as Pillow performs it."""
    H, W = img.shape[:2]
    new_w, new_h = size
    need_h, need_v = new_w != W, new_h != H
    cur = img
    y_first, y_last = 0, H
    if need_v:
        ksv, kkv, bv = coeffs(H, new_h, filt)
        y_first, y_last = int(bv[0, 0]), int(bv[new_h - 1, 0] + bv[new_h - 1, 1])
    if need_h:
        ksh, kkh, bh = coeffs(W, new_w, filt)
        rows = cur[y_first:y_last] if need_v else cur
        out = np.empty((rows.shape[0], new_w, 3), dtype=np.uint8)
        for xx in range(new_w):
            xmin, xmax = bh[xx]
            acc = (rows[:, xmin:xmin + xmax, :].astype(np.int64) * kkh[xx, :xmax].astype(np.int64)[None, :, None]).sum(axis=1)
            out[:, xx, :] = _clip8(acc + (1 << (PRECISION_BITS - 1)))
        cur = out
        if need_v:
            bv = bv.copy()
            bv[:, 0] -= y_first
    if need_v:
        out = np.empty((new_h, cur.shape[1], 3), dtype=np.uint8)
        for yy in range(new_h):
            ymin, ymax = bv[yy]
            acc = (cur[ymin:ymin + ymax].astype(np.int64) * kkv[yy, :ymax].astype(np.int64)[:, None, None]).sum(axis=0)
            out[yy] = _clip8(acc + (1 << (PRECISION_BITS - 1)))
        cur = out
    return cur.copy() if cur is img else cur


def pil_resize(img: np.ndarray, size, filt: int) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.fromarray(img).resize(size, Image.BILINEAR if filt == BILINEAR else Image.LANCZOS))


def to_tensor_normalize(img: np.ndarray) -> np.ndarray:
    """transforms.ToTensor() then Normalize((0.5,)*3, (0.5,)*3): HWC uint8 -> CHW float32"""
    t = img.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    return (t - np.float32(0.5)) / np.float32(0.5)


def grid_mask_array(grid: int, img_size: int) -> np.ndarray:
    """mask of pretrain.py:45-52 for the cells whose bit (i * 8 + j) of ``grid`` is CLEAR"""
    m = np.ones((3, img_size, img_size), dtype=np.float32)
    ps = img_size // 8
    for i in range(8):
        for j in range(8):
            if not (grid >> (i * 8 + j)) & 1:
                m[:, i * ps:(i + 1) * ps, j * ps:(j + 1) * ps] = 0
    return m


def draw_grid_mask(rng) -> int:
    """the 64 ``random.random() < 0.4`` draws of pretrain.py:47-50 (row-major) as a keep-bitmask"""
    grid = 0
    for i in range(8):
        for j in range(8):
            if not rng.random() < 0.4:
                grid |= 1 << (i * 8 + j)
    return grid


def dataset_item_ref(img: np.ndarray, grid: int, img_size: int = 256, resize=pil_resize):
    """MonetPhotoDataset.__getitem__ (pretrain.py:41-57) from a decoded RGB array -> (masked_image, image, mask)"""
    H, W = img.shape[:2]
    if W <= H:
        new_w, new_h = img_size, int(img_size * H / W)
    else:
        new_h, new_w = img_size, int(img_size * W / H)
    r = resize(img, (new_w, new_h), BILINEAR) if (new_w, new_h) != (W, H) else img
    top, left = int(round((new_h - img_size) / 2.0)), int(round((new_w - img_size) / 2.0))
    image = to_tensor_normalize(r[top:top + img_size, left:left + img_size])
    mask = grid_mask_array(grid, img_size)
    return image * mask, image, mask


def output_to_u8(y: np.ndarray) -> np.ndarray:
    """batch_process_images.py:213-217 on a (3, H, W) float32 array"""
    out = (y.astype(np.float32) + np.float32(1.0)) / np.float32(2.0)
    out = np.clip(out, 0, 1)
    return (out.transpose(1, 2, 0) * 255).astype(np.uint8)


def process_cyclegan_ref(model_fn, img: np.ndarray, target: int = 256, resize=pil_resize) -> np.ndarray:
    """process_cyclegan (batch_process_images.py:176-236) without file I/O; ``model_fn``: (1,3,T,T) float32 -> (1,3,T,T) float32"""
    height, width = img.shape[:2]
    if width > height:
        new_width, new_height = target, int(height * (target / width))
    else:
        new_height, new_width = target, int(width * (target / height))
    resized = resize(img, (new_width, new_height), LANCZOS)
    canvas = np.full((target, target, 3), 255, dtype=np.uint8)
    off_x, off_y = (target - new_width) // 2, (target - new_height) // 2
    canvas[off_y:off_y + new_height, off_x:off_x + new_width] = resized
    y = model_fn(to_tensor_normalize(canvas)[None])
    out = output_to_u8(np.asarray(y)[0])
    if width != height:
        aspect = width / height
        if aspect > 1:
            crop_w, crop_h = target, int(target / aspect)
        else:
            crop_h, crop_w = target, int(target * aspect)
        left, top = (target - crop_w) // 2, (target - crop_h) // 2
        out = out[top:top + crop_h, left:left + crop_w]
    if width * height <= 1024 * 1024:
        out = resize(np.ascontiguousarray(out), (width, height), LANCZOS)
    return out


def blend_simple(orig: np.ndarray, styled: np.ndarray, strength: float) -> np.ndarray:
    """batch_process_images.py:306-310 (mode 'simple'): the reference's own numpy expression on two uint8 HWC arrays"""
    result = orig * (1 - strength) + styled * strength
    return np.clip(result, 0, 255).astype(np.uint8)


def blend_weight_map(orig: np.ndarray, styled: np.ndarray, weight: np.ndarray) -> np.ndarray:
    """batch_process_images.py:340-342 + :352 (mode 'enhanced' with enhance_colors / smooth off; :386-387 is the same expression):
    weight (H, W) float64 -> per-pixel blend, clip, uint8"""
    weight = np.asarray(weight, dtype=float)[:, :, np.newaxis]
    result = orig * (1 - weight) + styled * weight
    return np.clip(result, 0, 255).astype(np.uint8)


def process_local_style_ref(model_fn, img: np.ndarray, mode="simple", strength=0.8, weight_map=None, target: int = 256,
                            resize=pil_resize) -> np.ndarray:
    """process_local_style (batch_process_images.py:255-441) without file I/O for mode 'simple', 'weight_map' (the 'enhanced'
    blend given its weight map) and the default branch; ``model_fn``: (1,3,T,T) float32 -> (1,3,T,T) float32"""
    height, width = img.shape[:2]
    if width > height:
        new_width, new_height = target, int(height * (target / width))
    else:
        new_height, new_width = target, int(width * (target / height))
    resized = resize(img, (new_width, new_height), LANCZOS)
    canvas = np.full((target, target, 3), 255, dtype=np.uint8)
    off_x, off_y = (target - new_width) // 2, (target - new_height) // 2
    canvas[off_y:off_y + new_height, off_x:off_x + new_width] = resized
    styled = output_to_u8(np.asarray(model_fn(to_tensor_normalize(canvas)[None]))[0])
    if mode == "simple":
        out = blend_simple(canvas, styled, strength)
    elif mode == "weight_map":
        out = blend_weight_map(canvas, styled, weight_map)
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
        out = out[top:top + crop_h, left:left + crop_w]
    if width * height <= 1024 * 1024:
        out = resize(np.ascontiguousarray(out), (width, height), LANCZOS)
    return out
