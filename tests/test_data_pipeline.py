"""datasets + transform chain of the reference driver (host side, CPU tests).
The Resize step is pinned to the half-pixel bilinear formula (oracle/data_ref.py); parity with torchvision itself is
unpinned because torchvision is not installed here (stated in both module headers)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import data_ref as DR
from osmosis_diffusion_code_amd.osmosis_utils import data as D


def test_natural_order():
    names = ["img10.png", "img2.png", "img1.png", "a_3_x.png", "a_12_x.png", "IMG1.png"]
    assert D.natsorted(names) == ["IMG1.png", "a_3_x.png", "a_12_x.png", "img1.png", "img2.png", "img10.png"]


@pytest.mark.parametrize("h,w", [(300, 400), (512, 256), (256, 256), (257, 300), (100, 180)])
def test_transform_chain_matches_oracle(h, w):
    rng = np.random.default_rng(h * 7 + w)
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    got = D.default_transform(256)(Image.fromarray(img))
    assert got.shape == (3, 256, 256) and got.dtype == torch.float32
    assert float(got.min()) >= -1.0 and float(got.max()) <= 1.0
    want = DR.transform(img, 256)
    # the source coordinate (dst + 0.5) * in/out - 0.5 is rounded in fp32 at magnitudes up to 512 (1 ulp = 6e-5) and
    # ATen may fuse the multiply-add; a 1-ulp coordinate change moves a weight by 6e-5 -> <= 1.2e-4 after Normalize
    assert np.allclose(got.numpy(), want, atol=2e-4), float(np.abs(got.numpy() - want).max())


def test_resize_rule_and_center_crop_offsets():
    x = torch.arange(3 * 5 * 7, dtype=torch.float32).view(3, 5, 7)
    assert D.resize(x, 10).shape == (3, 10, 14)            # smaller edge 5 -> 10, longer int(10 * 7 / 5)
    assert D.resize(x.transpose(1, 2), 10).shape == (3, 14, 10)
    assert D.resize(x, 5) is x                              # already that size: untouched
    c = D.center_crop(x, [2, 4])                            # top = round(1.5) = 2, left = round(1.5) = 2
    assert torch.equal(c, x[:, 2:4, 2:6])
    p = D.center_crop(torch.ones(1, 2, 2), [4, 4])          # smaller than the crop: zero padding
    assert p.shape == (1, 4, 4) and float(p.sum()) == 4.0 and float(p[0, 1:3, 1:3].sum()) == 4.0


def test_to_tensor_and_normalize():
    a = np.array([[[0, 128, 255]]], dtype=np.uint8)
    t = D.to_tensor(a)
    assert t.shape == (3, 1, 1) and torch.allclose(t.flatten(), torch.tensor([0.0, 128 / 255, 1.0]))
    assert torch.allclose(D.normalize(t).flatten(), torch.tensor([-1.0, 2 * 128 / 255 - 1, 1.0]), atol=1e-6)
    g = D.to_tensor(np.zeros((4, 5), dtype=np.uint8))
    assert g.shape == (1, 4, 5)


def test_images_folder_and_gt_datasets(tmp_path):
    rng = np.random.default_rng(0)
    root, rgb, dep = tmp_path / "in", tmp_path / "rgb", tmp_path / "depth"
    for d in (root, rgb, dep):
        os.makedirs(d)
    for name in ("s10.png", "s2.png"):
        Image.fromarray(rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)).save(root / name)
        Image.fromarray(rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)).save(rgb / name)
    Image.fromarray((rng.integers(0, 65536, (40, 60))).astype(np.uint16)).save(dep / "s2.png")    # 16-bit depth
    Image.fromarray(rng.integers(0, 256, (40, 60), dtype=np.uint8)).save(dep / "s10.png")         # 8-bit depth
    tf = D.default_transform(32)
    ds = D.ImagesFolder(str(root), tf)
    assert len(ds) == 2 and [ds[i][1] for i in range(2)] == ["s2.png", "s10.png"]
    assert ds[0][0].shape == (3, 32, 32)
    gt = D.ImagesFolder_GT(str(root), str(rgb), str(dep), tf)
    (img, g_rgb, g_dep), name = gt[0]
    assert name == "s2.png" and img.shape == g_rgb.shape == g_dep.shape == (3, 32, 32)
    assert torch.equal(g_dep[0], g_dep[1]) and torch.equal(g_dep[1], g_dep[2])        # depth replicated to RGB
    raw16 = np.asarray(Image.open(dep / "s2.png"))
    want = DR.transform(np.repeat((raw16 // 256).astype(np.uint8)[:, :, None], 3, axis=2), 32)
    assert np.allclose(g_dep.numpy(), want, atol=2e-4)
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False)))
    assert batch[0].shape == (2, 3, 32, 32) and list(batch[1]) == ["s2.png", "s10.png"]
