#!/usr/bin/env python3
"""GroupNorm pass timings at the UNet's shapes (HIP events, buffers rotated through > 256 MB so that no launch finds its
operands in the Infinity Cache by accident of the loop).  OSM_LIB selects a build (tools/ab-style comparisons).

    tools/gn_probe.py [--reps 20] [--shapes HW,C ...]
prints:  gn  <op>  HW,C   <us>   <GB/s of algorithmic bytes>
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from osmosis_diffusion_code_amd import ops  # noqa: E402

SHAPES = [(65536, 256), (65536, 512), (16384, 256), (16384, 512), (16384, 768), (4096, 256), (4096, 512), (4096, 768), (4096, 1024),
          (1024, 512), (1024, 1024), (1024, 1536), (256, 1024), (256, 2048), (64, 1024), (64, 2048)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--shapes", nargs="*", default=None)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--no-silu", action="store_true", help="normalise only (how much of a pass is the SiLU / dSiLU arithmetic?)")
    a = ap.parse_args()
    shapes = [tuple(int(v) for v in s.split(",")) for s in a.shapes] if a.shapes else SHAPES
    dev = torch.device("cuda:0")
    G, B = 32, a.batch
    for HW, C in shapes:
        nset = max(2, int(600e6 // (HW * C * 4 * 4)) + 1)
        nset = min(nset, 64)
        g = torch.Generator(device=dev).manual_seed(1)
        sets = []
        for _ in range(nset):
            x = torch.randn(B * HW, C, device=dev, generator=g)
            dy = torch.randn(B * HW, C, device=dev, generator=g)
            dx = torch.randn(B * HW, C, device=dev, generator=g)
            y = torch.empty(B * HW, C, device=dev)
            sets.append((x, dy, dx, y))
        gamma = torch.randn(C, device=dev, generator=g)
        beta = torch.randn(C, device=dev, generator=g)
        nch = ops.gn_nchunk(HW)
        part = torch.empty(B * nch * G * 2 + 1024, device=dev)
        stats = torch.empty(B * G * 2, device=dev)
        gstats = torch.empty(B * G * 2, device=dev)
        mx = torch.empty(B * ops.MAXABS_PARTS, device=dev, dtype=torch.int32)
        ops.gn_stats(ops.Mat.of(sets[0][0]), B, HW, G, part, stats)

        def fwd(s):
            x, dy, dx, y = s
            ops.gn_fwd(ops.Mat.of(x), ops.Mat.of(y), B, HW, G, part, stats, gamma, beta, maxabs=mx, silu=not a.no_silu)

        def apply(s):
            x, dy, dx, y = s
            ops.gn_apply(ops.Mat.of(x), ops.Mat.of(y), B, HW, G, stats, gamma, beta, maxabs=mx, silu=not a.no_silu)

        def bwd(s):
            x, dy, dx, y = s
            ops.gn_bwd(ops.Mat.of(x), ops.Mat.of(dy), ops.Mat.of(dx), B, HW, G, stats, gamma, beta, part, gstats,
                       addend=ops.Mat.of(dx), maxabs=mx, silu=not a.no_silu)

        def timed(name, fn, nb, note=""):
            for s in sets[:2]:
                fn(s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(a.reps):
                fn(sets[r % nset])
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.reps
            print(f"gn {name:9s} {HW},{C}  {us:8.1f} us  {nb * B * HW * C * 4 / us / 1e3:8.1f} GB/s {note}", flush=True)

        # GB/s: MINIMAL bytes of the operation (forward: read x, write y; backward: read x, dy, addend, write dx)
        for name, fn, nb in (("fwd", fwd, 2), ("apply", apply, 2), ("bwd", bwd, 4)):
            timed(name, fn, nb)


if __name__ == "__main__":
    main()
