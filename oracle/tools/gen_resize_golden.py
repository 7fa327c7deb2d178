#!/usr/bin/env python3
"""Golden vectors for the input transform chain of the reference driver (osmosis_sampling.py:46-49,
`ToTensor -> Resize(256) -> CenterCrop([256, 256]) -> Normalize(0.5, 0.5)`), SURVEY.md section 8(f) N3.

torchvision (pinned 0.14.1 by the reference's environment.yml) is NOT installed in this image, so the chain
itself cannot be executed here.  What this script pins instead is the ATen operator that torchvision's tensor
path dispatches to.  In torchvision 0.14.x, `transforms.Resize(256)` applied to a TENSOR runs (file
torchvision/transforms/functional.py `resize` -> `_compute_resized_output_size`, then
torchvision/transforms/functional_tensor.py `resize`; written from the published 0.14 sources, which are not
available offline to re-check):

    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)          # smaller edge -> size
    antialias = False if antialias is None else antialias          # tensors: NO antialiasing in 0.14
    img = torch.nn.functional.interpolate(img[None].float(), size=[new_h, new_w], mode="bilinear",
                                          align_corners=False, antialias=False)[0]

(`antialias=None` became "True with a warning" only in 0.15-0.17.)  The fixtures below are produced by exactly that
`interpolate` call of the torch in this container, on seeded uint8 images of awkward sizes, followed by
the CenterCrop / Normalize arithmetic, at size 256 (a 32 x 32 window and the sum of the output, to keep the file small) and at
size 32 (whole output).  tests/test_data_pipeline.py holds both `osmosis_utils/data.py` (product, torch) and
`oracle/data_ref.py` (independent numpy restatement of the half-pixel bilinear formula) to these vectors.

    python oracle/tools/gen_resize_golden.py        # rewrites tests/golden/resize_chain.npz
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.abspath(os.path.join(HERE, "..", "..", "tests", "golden", "resize_chain.npz"))
SIZES = [(117, 160), (160, 117), (96, 131), (163, 225), (64, 64), (70, 64)]   # (H, W); 163 x 225 = the 653 x 900 aspect of data/rgb_guidance


def chain(img_u8_hwc: np.ndarray, size: int) -> torch.Tensor:
    x = torch.from_numpy(img_u8_hwc).permute(2, 0, 1).to(torch.float32).div(255)          # ToTensor
    h, w = x.shape[-2:]
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    if (nh, nw) != (h, w):
        x = F.interpolate(x[None], size=[nh, nw], mode="bilinear", align_corners=False, antialias=False)[0]
    top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))             # CenterCrop([size, size])
    x = x[:, top:top + size, left:left + size]
    return (x - 0.5) / 0.5                                                                 # Normalize(0.5, 0.5)


def main():
    g = np.random.default_rng(20240807)
    out = {}
    for k, (h, w) in enumerate(SIZES):
        # smooth + noise so that interpolation errors are visible at every pixel
        yy, xx = np.mgrid[0:h, 0:w]
        base = 127 + 100 * np.sin(yy[..., None] / 9.0 + np.arange(3)) * np.cos(xx[..., None] / 13.0)
        img = np.clip(base + g.integers(-20, 21, size=(h, w, 3)), 0, 255).astype(np.uint8)
        out[f"img_{k}"] = img
        out[f"out32_{k}"] = chain(img, 32).numpy()
        full = chain(img, 256)
        out[f"out256_shape_{k}"] = np.asarray(full.shape)
        out[f"out256_win_{k}"] = full[:, 112:144, 112:144].numpy().copy()
        out[f"out256_sum_{k}"] = np.asarray(full.double().sum().item())
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
