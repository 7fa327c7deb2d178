#!/usr/bin/env python3
"""MFMA-pipe utilisation of the dominant kernel from rocprofv3 PMC counters (counters only, no traces):

    tools/mfma_busy.py <out.json>          # runs rocprofv3 --pmc over tools/conv_probe.py itself (on a GPU box)

Counters: SQ_VALU_MFMA_BUSY_CYCLES (cycles the matrix pipe is busy, summed over every SIMD of the chip: 32 per
v_mfma_f32_32x32x16_bf16 -- the count below reproduces the analytic number of MFMAs exactly), GRBM_GUI_ACTIVE
(shader-clock cycles the GPU was busy during the dispatch; rocprofv3 reports the SUM over the 8 XCDs, each with its
own GRBM, hence the / 8), SQ_WAVE_CYCLES, SQ_BUSY_CU_CYCLES.  Derived per dispatch:
    gui_cycles      = GRBM_GUI_ACTIVE / 8
    mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x gui_cycles)
    clock_ghz       = gui_cycles / dispatch duration               (what the power manager grants under this load)
    expected_mfma   = algorithmic flops x N / (2 x 32 x 32 x 16)   (N MFMAs per fp32 product group: 6 for bf16x6, 3 for f16x3;
                      x 16/36 for the Winograd F(2x2,3x3) kernel, which multiplies 16 instead of 36 times per tile)
    mfma_count_seen = SQ_VALU_MFMA_BUSY_CYCLES / 32                (sanity check against expected_mfma)
Each conv shape is measured three times: the direct halo-tile kernel (bf16x6), the Winograd kernel in bf16x6 and in f16x3 (the
arithmetic the engine uses by default); then the three kernels of the flash-attention family at the three block shapes of the
UNet (tools/attn_probe.py).
"""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = ["1,256,256,256,256,3", "1,128,128,512,512,3", "1,64,64,512,512,3"]
ATTN = ["1,1024,8", "1,256,16", "1,64,16"]
COUNTERS = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES"]
# (label, conv_probe arguments, kernel-name substring, MFMAs per fp32 product, executed fraction of the algorithmic products)
CONV_VARIANTS = [
    ("conv3_halo_bf16s_kernel<3,*> (direct, bf16x6)", ["--mode", "bf16x6"], "conv3_halo", 6, 1.0),
    ("conv3_wino8_kernel<3,*,false> (Winograd F(2x2,3x3), bf16x6)", ["--mode", "bf16x6", "--winograd"], "conv3_wino8", 6, 16.0 / 36.0),
    ("conv3_wino8_kernel<2,*,true> (Winograd F(2x2,3x3), f16x3: the default)", ["--mode", "f16x3"], "conv3_wino8", 3, 16.0 / 36.0),
]


# the dominant kernel of BASELINE config 5 (B = 32, use_fp16): the direct kernel, one fp16 MFMA per product
F16_SHAPES = ["32,256,256,256,256,3", "32,128,128,256,256,3", "32,64,64,512,512,3"]
F16_VARIANT = ("conv3_halo_bf16s_kernel<1,*> (direct, fp16 storage: config 5)", ["--mode", "f16"], "conv3_halo", 1, 1.0)


def collect(cmd_tail, substrs):
    """One rocprofv3 --pmc run (counters only); per kernel-name substring: counter -> [sum, dispatches, total us]."""
    d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc", *COUNTERS, "--output-format", "csv", "-d", d, "--", sys.executable] + cmd_tail
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                   timeout=400)
    accs = {k: {} for k in substrs}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for k in substrs:
                if k in r["Kernel_Name"]:
                    a = accs[k].setdefault(r["Counter_Name"], [0.0, 0, 0.0])
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
                    a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return accs


def row(label, acc, flops, nmfma, exec_frac, extra):
    if "GRBM_GUI_ACTIVE" not in acc:
        return dict(extra, kernel=label, error="no counters collected")
    avg = {k: v[0] / v[1] for k, v in acc.items()}
    us = acc["GRBM_GUI_ACTIVE"][2] / acc["GRBM_GUI_ACTIVE"][1]
    gui = avg["GRBM_GUI_ACTIVE"] / 8.0
    out = dict(extra, kernel=label, dispatches=acc["GRBM_GUI_ACTIVE"][1], avg_us_under_pmc=round(us, 2),
               counters_avg_per_dispatch={k: round(v, 1) for k, v in avg.items()},
               mfma_busy_frac=round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui), 4), clock_ghz=round(gui / (us * 1e3), 3),
               mfma_instructions_seen=round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 32))
    if flops is not None:
        out.update(expected_mfma_instructions=round(flops * exec_frac * nmfma / (2 * 32 * 32 * 16)),
                   algorithmic_tflops_under_pmc=round(flops / us / 1e6, 1),
                   roof_at_that_clock_tflops=round(2500.0 / nmfma * (gui / (us * 1e3)) / 2.4, 1))
    return out


def main():
    out_json = sys.argv[1]
    rows = []
    probe = os.path.join(REPO, "tools", "conv_probe.py")
    for shape in SHAPES:
        B, H, W, Cin, Cout, k = (int(v) for v in shape.split(","))
        flops = 2.0 * B * H * W * Cin * Cout * k * k
        for label, args, sub, nmfma, ef in CONV_VARIANTS:
            acc = collect([probe, "--shape", shape, "--iters", "10"] + args, [sub])[sub]
            rows.append(row(label, acc, flops, nmfma, ef, {"shape_B,H,W,Cin,Cout,k": shape}))
    for shape in F16_SHAPES:
        B, H, W, Cin, Cout, k = (int(v) for v in shape.split(","))
        label, args, sub, nmfma, ef = F16_VARIANT
        acc = collect([probe, "--shape", shape, "--iters", "6"] + args, [sub])[sub]
        rows.append(row(label, acc, 2.0 * B * H * W * Cin * Cout * k * k, nmfma, ef, {"shape_B,H,W,Cin,Cout,k": shape}))
    # attention cores (a counter behind bench.py's attention_mfma_util): per kernel of the flash family
    for shape in ATTN:
        B, T, heads = (int(v) for v in shape.split(","))
        per = 2.0 * B * heads * T * T * 64
        # round 6: the product's cores are the f16x3 instances (three MFMAs per product); the bf16x6 ones (OSM_ATTN_F16X3=0) beside them
        for arith, suffix, nm in (("f16x3", "_hp_kernel", 3), ("bf16x6", "_kernel", 6)):
            subs = ["flash_fwd" + suffix, "flash_bwd_q" + suffix, "flash_bwd_kv" + suffix]
            accs = collect([os.path.join(REPO, "tools", "attn_probe.py"), "--shape", shape, "--iters", "10", "--arith", arith], subs)
            for sub, ngemm in zip(subs, (2, 3, 4)):
                # forward: S, PV; bwd_q: S, dP, dq; bwd_kv: S, dP, dv, dk (S and dP are recomputed by both backward kernels)
                rows.append(row(f"{sub} ({arith})", accs[sub], per * ngemm, nm, 1.0, {"shape_B,T,heads": shape, "gemms_executed": ngemm}))
    json.dump({"note": __doc__.split("Counters:")[1].strip(), "rows": rows}, open(out_json, "w"), indent=1)
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
