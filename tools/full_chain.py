#!/usr/bin/env python3
"""One complete image through the guided sampler (BASELINE config 1: osmosis_sample_config.yaml, T = 1000,
revised underwater operator, n_iter = 20) with the per-image driver of sampling.py; seeded synthetic weights and
measurement.  Prints wall time, steps/s over the WHOLE chain (frozen-phi and phi-update regimes) and output sanity.

    python tools/full_chain.py [--steps 1000] [--batch 1]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (configuration constants of the benchmark)
from osmosis_diffusion_code_amd import sampling  # noqa: E402
from osmosis_diffusion_code_amd.guided_diffusion import unet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--last", type=int, default=0,
                    help="run only the last N steps of the chain from a small-variance x (seeded synthetic weights do not "
                         "denoise, so the x0 prediction of a full chain leaves the operator's range early: SURVEY F10)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = unet.create_model(**bench.UNET_KW)
    model.reset_parameters(1234)
    model = model.to(dev).eval()
    diffusion = dict(bench.DIFFUSION)
    if a.steps != 1000:
        diffusion["timestep_respacing"] = str(a.steps)
    cfg = dict(measurement=dict(operator=dict(name="underwater_physical_revised", **bench.OPERATOR),
                                noise=dict(name="clean")),
               conditioning=dict(method="osmosis", params=bench.COND), sample_pattern=bench.PATTERN,
               aux_loss=dict(aux_loss=bench.AUX), diffusion=diffusion,
               unet_model=dict(pretrain_model="osmosis"), manual_seed=0, degamma_input=False, rgb_guidance=False)
    _, y = bench.synthetic_inputs(0, a.batch, 256)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kw = {}
    if a.last:
        kw = dict(index_range=(a.last - 1, 0), x_scale=0.1)
    res = sampling.restore_image(model, y.to(dev), cfg, **kw)[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = a.last or a.steps
    out = {"steps": n, "batch": a.batch, "seconds": round(dt, 2),
           "denoise_steps_per_sec": round(n * a.batch / dt, 2),
           "finite": bool(torch.isfinite(res["pred_xstart"]).all() and torch.isfinite(res["sample"]).all()),
           "final_loss": [float(v) for v in res["loss"].ravel()],
           "phi_inf": [round(float(v), 4) for v in res["phi"]["phi_inf"].ravel()],
           "norm_loss_final": res["norm_loss_final"]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
