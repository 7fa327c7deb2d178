#!/usr/bin/env python3
"""Same-box, same-process A/B of ENGINE-level switches on the whole guided step (BASELINE config 2 by default): the model and
its weight images are built once; every variant gets a fresh engine (launch plans + graphs are re-recorded under the
variant's environment / knobs) and the variants are timed round-robin, so box-to-box and minute-to-minute clock drift hits
them alike.  Per-kernel sums overstate what a change buys (DESIGN learned 34): this is the number that decides.

    tools/step_ab.py [--steps 10] [--rounds 4] [--batch 1] base "nostats:OSM_FUSE_STATS=0" ...
a variant is  name[:k=v,...]  where k is an environment variable (read when the engine is built).
Variables that change the WEIGHT IMAGES (OSM_WINOGRAD, OSM_CONV_MODE, OSM_F16X3_1X1) need a process of their own."""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--conv-mode", default="f16x3")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 5])
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(tiny=False, image_size=256, batch=a.batch, conv_mode=a.conv_mode)
    if a.config == 2:
        model, sampler, cond = bench.build_case(args, dev, a.batch, conv_mode=a.conv_mode)
    else:
        c = bench.SECONDARY[0 if a.config == 3 else 1]
        a.batch = args.batch = c["batch"] if a.batch == 1 else a.batch
        model, sampler, cond = bench.build_case(args, dev, a.batch, c["unet"], c["diffusion"], c["operator"], c["cond"], c["aux"],
                                                conv_mode=a.conv_mode)
    variants = []
    for v in a.variants:
        name, _, kv = v.partition(":")
        env = {}
        for item in filter(None, kv.split(",")):
            k, _, val = item.partition("=")
            env[k] = val
        variants.append((name, env))
    base_env = dict(os.environ)
    times = {n: [] for n, _ in variants}
    for r in range(a.rounds):
        for name, env in variants:
            os.environ.clear()
            os.environ.update(base_env)
            os.environ.update(env)
            model._engines = {}
            torch.cuda.empty_cache()
            dt, finite = bench.timed_steps(args, dev, model, sampler, cond, a.batch, 0, a.steps, 3)
            times[name].append(1e3 * dt / a.steps)
            print(f"round {r} {name:24s} {1e3 * dt / a.steps:8.3f} ms/step  finite={finite}", flush=True)
    base = statistics.median(times[variants[0][0]])
    for name, _ in variants:
        m = statistics.median(times[name])
        print(f"{name:24s} median {m:8.3f} ms/step  min {min(times[name]):8.3f}  vs {variants[0][0]}: {m - base:+.3f} ms", flush=True)


if __name__ == "__main__":
    main()
