"""ORACLE (test infrastructure only).  numpy restatement of the input transform chain of the reference driver
(osmosis_sampling.py:46-49) as the pinned torchvision 0.14.1 executes it on tensors:
  ToTensor (uint8 HWC / 255 -> CHW float32), Resize(256) = bilinear interpolation with half-pixel centres
  (src = (dst + 0.5) * in/out - 0.5, clamped at 0; no antialiasing), smaller edge -> 256 and the longer edge
  int(256 * long / short), CenterCrop (offset round((H - h) / 2)), Normalize ((x - 0.5) / 0.5).
`dtype=np.float64` evaluates the same published formula in double: that evaluation -- independent of torch -- generates
tests/golden/resize_chain.npz (oracle/tools/gen_resize_golden.py), to which the product chain, a direct ATen
`interpolate(..., "bilinear", align_corners=False, antialias=False)` call and the fp32 evaluation below are all held.
torchvision itself is not installed in the build container: the torchvision-0.14.1 wrapper stays un-pinned.
"""
import numpy as np


def to_tensor(hwc_uint8, dtype=np.float32):
    a = np.asarray(hwc_uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return (a.astype(dtype) / dtype(255)).transpose(2, 0, 1)


def _axis_weights(n_in, n_out, dtype=np.float32):
    scale = dtype(n_in) / dtype(n_out)
    src = (np.arange(n_out, dtype=dtype) + dtype(0.5)) * scale - dtype(0.5)
    src = np.maximum(src, 0)
    i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    w1 = (src - i0.astype(dtype)).astype(dtype)
    return i0, i1, dtype(1) - w1, w1


def resize_short(chw, size):
    dtype = chw.dtype.type
    c, h, w = chw.shape
    if w <= h:
        nw, nh = size, int(size * h / w)
    else:
        nh, nw = size, int(size * w / h)
    if (nh, nw) == (h, w):
        return chw
    y0, y1, wy0, wy1 = _axis_weights(h, nh, dtype)
    x0, x1, wx0, wx1 = _axis_weights(w, nw, dtype)
    rows = chw[:, y0, :] * wy0[None, :, None] + chw[:, y1, :] * wy1[None, :, None]
    return (rows[:, :, x0] * wx0[None, None, :] + rows[:, :, x1] * wx1[None, None, :]).astype(dtype)


def center_crop(chw, th, tw):
    c, h, w = chw.shape
    top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
    return chw[:, top:top + th, left:left + tw]


def transform(hwc_uint8, size=256, dtype=np.float32):
    x = center_crop(resize_short(to_tensor(hwc_uint8, dtype), size), size, size)
    return (x - dtype(0.5)) / dtype(0.5)
