#!/usr/bin/env python3
"""Does the FREE-RUNNING guided chain agree HIP vs oracle end to end when the network is contractive?  The seeded test network is
not (tests/test_drift_gpu.py: the oracle diverges from its own 1e-7-perturbed copy); a trained one is.  Here the output head of the
seeded tiny network is damped by a factor (a freshly constructed reference UNet has a ZERO head: unet.py `zero_module`), and the
300-step chain is run free on both sides:   tools/contractive_probe.py [factors ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_drift_gpu as T  # noqa: E402


def main():
    factors = [float(v) for v in sys.argv[1:]] or [1.0, 0.1, 0.02, 0.0]
    n, size = 300, 64
    for f in factors:
        cfg, sd, sampler, tb, x_T, y, noise = T.setup(T.TINY_KW, size, n, 21)
        sd = dict(sd)
        for k in [k for k in sd if k.startswith("out.") and k.endswith((".weight", ".bias")) and sd[k].dim() in (1, 4) and k.split(".")[1] == "2"]:
            sd[k] = sd[k] * f
        ref = T.oracle_chain(cfg, sd, tb, x_T, y, noise, 8)
        bump = 1e-7 * torch.randn(x_T.shape, generator=torch.Generator().manual_seed(99))
        pert = T.oracle_chain(cfg, sd, tb, x_T + bump, y, noise, 8)
        model = T.hip_model(T.TINY_KW, sd)
        free = T.diff_curve(T.hip_chain(model, sampler, x_T, y, noise), ref, n)
        self_ = T.diff_curve(pert, ref, n)
        ef, es = T.envelope(free), T.envelope(self_)
        fin = free[-1]
        print(f"head x {f:g}: HIP vs oracle after 1/10/100/300 steps {ef[0]:.1e} {ef[9]:.1e} {ef[99]:.1e} {ef[-1]:.1e} | oracle vs itself "
              f"{es[0]:.1e} {es[9]:.1e} {es[99]:.1e} {es[-1]:.1e} | final x0 err {fin['x0']:.2e} (|x0| {fin['x0_absmax_ref']:.2f}) phi {fin['phi']:.1e} "
              f"loss {fin['loss']:.4f} vs {fin['loss_ref']:.4f}", flush=True)
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
