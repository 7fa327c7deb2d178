#!/usr/bin/env python3
"""Committed ATen vectors for the input transform chain: the SAME images as tests/golden/resize_chain.npz (the
float64 restatement) taken through the operator call torchvision 0.14.x makes for tensor inputs,
`torch.nn.functional.interpolate(x[None], size=[nh, nw], mode="bilinear", align_corners=False, antialias=False)`, then
CenterCrop / Normalize -- written against torch directly (not through the product code), run ONCE here and committed, so that
the product chain is held bit for bit to vectors that do not move with the code under test.  The provenance (torch version,
CPU build) is stored in the file; a different torch build may differ in the last bit (the test then reports which).

    python oracle/tools/gen_resize_aten_golden.py        # rewrites tests/golden/resize_chain_aten.npz
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.abspath(os.path.join(HERE, "..", "..", "tests", "golden", "resize_chain.npz"))
OUT = os.path.abspath(os.path.join(HERE, "..", "..", "tests", "golden", "resize_chain_aten.npz"))


def aten_chain(img_u8_hwc, size):
    x = torch.from_numpy(img_u8_hwc).permute(2, 0, 1).to(torch.float32).div(255)
    h, w = x.shape[-2:]
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    if (nh, nw) != (h, w):
        x = F.interpolate(x[None], size=[nh, nw], mode="bilinear", align_corners=False, antialias=False)[0]
    top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))
    x = x[:, top:top + size, left:left + size]
    return ((x - 0.5) / 0.5).numpy()


def main():
    g = np.load(GOLD)
    out = {"generator": np.array(f"oracle/tools/gen_resize_aten_golden.py: ATen interpolate (bilinear, align_corners=False, "
                                 f"antialias=False), torch {torch.__version__}, CPU; inputs = img_k of resize_chain.npz")}
    k = 0
    while f"img_{k}" in g:
        img = g[f"img_{k}"]
        out[f"out32_{k}"] = aten_chain(img, 32)
        out[f"out256_win_{k}"] = aten_chain(img, 256)[:, 112:144, 112:144].copy()
        k += 1
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, k, "images", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
