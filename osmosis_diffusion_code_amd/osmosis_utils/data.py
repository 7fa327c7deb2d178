"""Input pipeline of the reference driver: `osmosis_utils/data.py:15-109`
(`ImagesFolder`, `ImagesFolder_GT`) and the transform chain of `osmosis_sampling.py:46-49`
    ToTensor -> Resize(256) -> CenterCrop([256, 256]) -> Normalize(0.5, 0.5)
without torchvision / cv2 / natsort (none of them is in this image): PIL + torch only.

Resize follows the pinned torchvision 0.14.1 behaviour for TENSOR inputs (the chain resizes after ToTensor):
bilinear, align_corners=False, NO antialiasing, smaller edge -> `size`, longer edge -> int(size * long / short).
torchvision is absent here, so the pin is one level down: tests/golden/resize_chain.npz holds the output of the
ATen call torchvision 0.14.x makes (`interpolate(..., "bilinear", align_corners=False, antialias=False)`,
oracle/tools/gen_resize_golden.py); this module and the numpy restatement oracle/data_ref.py are both tested against it.
Host-side code; the sampler takes the resulting [B,3,256,256] tensor in [-1, 1].
"""
import glob
import os
import re
from os.path import join as pjoin

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image
from torch.utils.data import Dataset


# ----------------------------------------------------------------------------- natural ordering (natsort.natsorted)
def _natural_key(s):
    return [(0, int(p), "") if p.isdigit() else (1, 0, p) for p in re.split(r"(\d+)", str(s)) if p != ""]


def natsorted(items):
    """Default natsort order: digit runs compare as unsigned integers, the rest as text ('img2' < 'img10')."""
    return sorted(items, key=_natural_key)


# ----------------------------------------------------------------------------- transforms
def to_tensor(pic):
    """transforms.ToTensor: PIL image / HxWxC uint8 array -> float32 CxHxW in [0, 1] (other dtypes: no scaling)."""
    arr = np.array(pic)          # a writable copy (PIL buffers are read-only)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)
    if t.dtype == torch.uint8:
        return t.to(torch.float32).div(255)
    return t.to(torch.float32) if t.dtype != torch.float32 else t


def resize(img, size):
    """transforms.Resize(size=int) on a tensor [..., H, W]: smaller edge -> size, bilinear, no antialias."""
    h, w = img.shape[-2:]
    if isinstance(size, int):
        short, long_ = (w, h) if w <= h else (h, w)
        new_short, new_long = size, int(size * long_ / short)
        nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    else:
        nh, nw = size
    if (nh, nw) == (h, w):
        return img
    x = img if img.dim() == 4 else img.unsqueeze(0)
    y = F.interpolate(x.to(torch.float32), size=(nh, nw), mode="bilinear", align_corners=False, antialias=False)
    return y if img.dim() == 4 else y.squeeze(0)


def center_crop(img, output_size):
    """transforms.CenterCrop: zero-pads when the image is smaller, then crops at round((H - h) / 2)."""
    th, tw = (output_size, output_size) if isinstance(output_size, int) else output_size
    h, w = img.shape[-2:]
    if tw > w or th > h:
        pl = (tw - w) // 2 if tw > w else 0
        pt = (th - h) // 2 if th > h else 0
        pr = (tw - w + 1) // 2 if tw > w else 0
        pb = (th - h + 1) // 2 if th > h else 0
        img = F.pad(img, (pl, pr, pt, pb))
        h, w = img.shape[-2:]
        if (th, tw) == (h, w):
            return img
    top = int(round((h - th) / 2.0))
    left = int(round((w - tw) / 2.0))
    return img[..., top:top + th, left:left + tw]


def normalize(img, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    m = torch.as_tensor(mean, dtype=img.dtype).view(-1, 1, 1)
    s = torch.as_tensor(std, dtype=img.dtype).view(-1, 1, 1)
    return (img - m) / s


def default_transform(size=256):
    """osmosis_sampling.py:46-49."""
    def apply(pic):
        return normalize(center_crop(resize(to_tensor(pic), size), [size, size]))
    return apply


# ----------------------------------------------------------------------------- datasets
class ImagesFolder(Dataset):
    """data.py:15-38: every file of `root_dir` in natural order -> (transformed image, file name)."""

    def __init__(self, root_dir, transform=None):
        self.root_dir = root_dir
        self.images_list = natsorted(os.listdir(root_dir))
        self.transform = transform

    def __len__(self):
        return len(self.images_list)

    def __getitem__(self, idx):
        image = Image.open(os.path.join(self.root_dir, self.images_list[idx]))
        if self.transform is not None:
            image = self.transform(image)
        return image, self.images_list[idx]


class ImagesFolder_GT(Dataset):
    """data.py:73-109 (simulation config): ([image, gt_rgb, gt_depth as 3 equal channels], image file name).
    16-bit depth maps are reduced to 8 bits by an integer division by 256, like the reference."""

    def __init__(self, root_dir, gt_rgb_dir, gt_depth_dir, transform=None):
        self.gt_rgb_dir, self.gt_depth_dir, self.root_dir = gt_rgb_dir, gt_depth_dir, root_dir
        self.gt_rgb_list = natsorted(glob.glob(pjoin(gt_rgb_dir, "*.*")))
        self.gt_depth_list = natsorted(glob.glob(pjoin(gt_depth_dir, "*.*")))
        self.images_list = natsorted(glob.glob(pjoin(root_dir, "*.*")))
        self.transform = transform

    def __len__(self):
        return len(self.gt_rgb_list)

    def __getitem__(self, idx):
        image_name = os.path.basename(self.images_list[idx])
        image = Image.open(self.images_list[idx])
        gt_rgb = Image.open(self.gt_rgb_list[idx])
        depth = np.asarray(Image.open(self.gt_depth_list[idx]))
        if depth.dtype == np.uint16 or depth.dtype == np.int32:      # PIL opens 16-bit PNGs as I;16 / I
            depth = (depth // 256).astype(np.uint8)
        gt_depth = Image.fromarray(depth)
        if self.transform is not None:
            image = self.transform(image)
            gt_rgb = self.transform(gt_rgb)
            gt_depth = self.transform(gt_depth.convert(mode="RGB"))
        return [image, gt_rgb, gt_depth], image_name
