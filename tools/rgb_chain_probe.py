#!/usr/bin/env python3
"""A few steps of the reference's SHIPPED rgb-guidance configuration (configs/rgb_guidance_sample_config.yaml: ddpm, `ps` conditioning on
the identity operator, clip_denoised) at full size through sampling.restore_image -- the workload of bench.py's `rgb_guidance_chain` leg,
cut short for a kernel trace:

    cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d out -- python tools/rgb_chain_probe.py --steps 40

Prints one JSON line (ms per step, finite, fused)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    from osmosis_diffusion_code_amd import sampling
    from osmosis_diffusion_code_amd.guided_diffusion import gaussian_diffusion as gd
    dev = torch.device("cuda:0")
    args = argparse.Namespace(tiny=False, image_size=256, batch=1, conv_mode="f16x3")
    model, _, _ = bench.build_case(args, dev, 1, conv_mode="f16x3")
    _, y = bench.synthetic_inputs(0, 1, 256)
    y = y.to(dev)
    fell_back = []
    orig = gd.GaussianDiffusion._generic_loop
    gd.GaussianDiffusion._generic_loop = lambda self, *x, **k: (fell_back.append(1), orig(self, *x, **k))[1]
    T = 1000
    sampling.restore_image(model, y, bench.RGB_GUIDANCE, noise_seed=0, index_range=(T - 1, T - a.warmup))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = sampling.restore_image(model, y, bench.RGB_GUIDANCE, noise_seed=0, index_range=(T - 1, T - a.steps))[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "rgb_guidance_sample_config.yaml, B = 1, 256 x 256", "steps": a.steps, "ms_per_step": round(1e3 * dt / a.steps, 3),
                      "fused_loop": not fell_back, "finite": bool(torch.isfinite(res["sample"]).all())}))


if __name__ == "__main__":
    main()
