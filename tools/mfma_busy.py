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
    expected_mfma   = algorithmic flops x 6 / (2 x 32 x 32 x 16)   (bf16x6: six MFMAs per fp32 product group;
                      x 16/36 for the Winograd F(2x2,3x3) kernel, which multiplies 16 instead of 36 times per tile)
    mfma_count_seen = SQ_VALU_MFMA_BUSY_CYCLES / 32                (sanity check against expected_mfma)
Each shape is measured twice: the direct halo-tile kernel and (--winograd) the Winograd kernel the engine now uses.
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
COUNTERS = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES"]


def main():
    out_json = sys.argv[1]
    rows = []
    for shape, wino in [(s, w) for s in SHAPES for w in (False, True)]:
        d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        cmd = ["rocprofv3", "--pmc", *COUNTERS, "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.join(REPO, "tools", "conv_probe.py"), "--shape", shape, "--iters", "10"] + (["--winograd"] if wino else [])
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=400)
        acc = {}
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if ("conv3_wino" if wino else "conv3_halo") not in r["Kernel_Name"]:
                    continue
                a = acc.setdefault(r["Counter_Name"], [0.0, 0, 0.0])
                a[0] += float(r["Counter_Value"])
                a[1] += 1
                a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        if "GRBM_GUI_ACTIVE" not in acc:
            rows.append({"shape": shape, "error": "no counters collected"})
            continue
        avg = {k: v[0] / v[1] for k, v in acc.items()}
        us = acc["GRBM_GUI_ACTIVE"][2] / acc["GRBM_GUI_ACTIVE"][1]
        B, H, W, Cin, Cout, k = (int(v) for v in shape.split(","))
        flops = 2.0 * B * H * W * Cin * Cout * k * k
        exec_frac = 16.0 / 36.0 if wino else 1.0
        rows.append({
            "shape_B,H,W,Cin,Cout,k": shape,
            "kernel": "conv3_wino_kernel<3,*> (bf16x6, Winograd F(2x2,3x3))" if wino else "conv3_halo_bf16s_kernel<3,*> (bf16x6)", "dispatches": acc["GRBM_GUI_ACTIVE"][1],
            "avg_us_under_pmc": round(us, 2), "counters_avg_per_dispatch": {k: round(v, 1) for k, v in avg.items()},
            "mfma_busy_frac": round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * avg["GRBM_GUI_ACTIVE"] / 8.0), 4),
            "clock_ghz": round(avg["GRBM_GUI_ACTIVE"] / 8.0 / (us * 1e3), 3),
            "bf16x6_roof_at_that_clock_tflops": round(2500.0 / 6.0 * (avg["GRBM_GUI_ACTIVE"] / 8.0 / (us * 1e3)) / 2.4, 1),
            "expected_mfma_instructions": round(flops * exec_frac * 6 / (2 * 32 * 32 * 16)),
            "mfma_instructions_seen": round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 32),
            "algorithmic_tflops_under_pmc": round(flops / us / 1e6, 1)})
    json.dump({"note": __doc__.split("Counters:")[1].strip(), "rows": rows}, open(out_json, "w"), indent=1)
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
