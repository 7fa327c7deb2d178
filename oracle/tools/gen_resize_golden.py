#!/usr/bin/env python3
"""Golden vectors for the input transform chain of the reference driver (osmosis_sampling.py:46-49,
`ToTensor -> Resize(256) -> CenterCrop([256, 256]) -> Normalize(0.5, 0.5)`), SURVEY.md section 8(f) N3.

torchvision (pinned 0.14.1 by the reference's environment.yml) is NOT installed in this image, so the chain itself cannot
be executed here.  In torchvision 0.14.x, `transforms.Resize(256)` applied to a TENSOR runs (torchvision/transforms/
functional.py `resize` -> `_compute_resized_output_size`, then functional_tensor.py `resize`; written from the published
0.14 sources, which are not available offline to re-check):

    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)          # smaller edge -> size
    antialias = False if antialias is None else antialias          # tensors: NO antialiasing in 0.14
    img = torch.nn.functional.interpolate(img[None].float(), size=[new_h, new_w], mode="bilinear",
                                          align_corners=False, antialias=False)[0]

(`antialias=None` became "True with a warning" only in 0.15-0.17.)  Round 3: the vectors are produced by the numpy
restatement of that chain evaluated in FLOAT64 (oracle/data_ref.py, `dtype=np.float64`: half-pixel source coordinate
(dst + 0.5) * in/out - 0.5 clamped at 0, two-tap linear weights, CenterCrop offset round((H - h) / 2), Normalize) -- no torch
call is involved, so the golden no longer pins torch against itself.  tests/test_data_pipeline.py holds the product chain
(osmosis_utils/data.py), a direct ATen `interpolate` call and the fp32 numpy evaluation to these vectors; the fp32
evaluations differ from the float64 one by <= 1.2e-4 (the source coordinate is rounded in fp32 at magnitudes up to 512).

    python oracle/tools/gen_resize_golden.py        # rewrites tests/golden/resize_chain.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import data_ref as DR  # noqa: E402

OUT = os.path.abspath(os.path.join(HERE, "..", "..", "tests", "golden", "resize_chain.npz"))
SIZES = [(117, 160), (160, 117), (96, 131), (163, 225), (64, 64), (70, 64)]   # (H, W); 163 x 225 = the 653 x 900 aspect of data/rgb_guidance


def main():
    g = np.random.default_rng(20240807)
    out = {}
    for k, (h, w) in enumerate(SIZES):
        # smooth + noise so that interpolation errors are visible at every pixel
        yy, xx = np.mgrid[0:h, 0:w]
        base = 127 + 100 * np.sin(yy[..., None] / 9.0 + np.arange(3)) * np.cos(xx[..., None] / 13.0)
        img = np.clip(base + g.integers(-20, 21, size=(h, w, 3)), 0, 255).astype(np.uint8)
        out[f"img_{k}"] = img
        out[f"out32_{k}"] = DR.transform(img, 32, np.float64)
        full = DR.transform(img, 256, np.float64)
        out[f"out256_shape_{k}"] = np.asarray(full.shape)
        out[f"out256_win_{k}"] = full[:, 112:144, 112:144].copy()
        out[f"out256_sum_{k}"] = np.asarray(full.sum())
    out["generator"] = np.asarray("oracle/data_ref.py transform(dtype=float64): numpy only, no torch")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
