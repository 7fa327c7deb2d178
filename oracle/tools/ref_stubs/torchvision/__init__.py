"""Import-time stand-in for torchvision, used ONLY by oracle/tools/gen_golden.py in the
build container so that /root/reference's hot-path modules import.  None of these
symbols is touched on the numeric path (record=False, save_grids_path=None).
The reference does `from torchvision import torch` (guided_diffusion/measurements.py:8)."""
import torch  # noqa: F401
from . import utils, transforms  # noqa: F401
