#!/usr/bin/env python3
"""Time the inner phi loop of one guided step (osm_phys_optimize: 20 x (reduce, finalize) + gradient) at 256 x 256 through the
conditioning method, HIP events:   tools/phys_probe.py [--iters 50]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from osmosis_diffusion_code_amd.guided_diffusion import condition_methods as CM  # noqa: E402
from osmosis_diffusion_code_amd.guided_diffusion import measurements as M  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    dev = "cuda:0"
    op = M.get_operator("underwater_physical_revised", device=dev, batch_size=a.batch, **bench.OPERATOR)
    cond = CM.get_conditioning_method("osmosis", op, M.get_noise("clean"), **bench.COND, **bench.PATTERN, aux_loss=bench.AUX)
    g = torch.Generator(device=dev).manual_seed(0)
    x0 = 0.6 * torch.randn(a.batch, 4, 256, 256, device=dev, generator=g)
    y = torch.rand(a.batch, 3, 256, 256, device=dev, generator=g) * 1.6 - 0.8
    for _ in range(5):
        cond.loss_grad_x0(x0, y, freeze_phi=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        cond.loss_grad_x0(x0, y, freeze_phi=False)
    e1.record()
    torch.cuda.synchronize()
    print(f"phi loop (n_iter = {cond.n_iter}, B = {a.batch}, 256 x 256): {e0.elapsed_time(e1) / a.iters * 1e3:.1f} us per guided step", flush=True)


if __name__ == "__main__":
    main()
