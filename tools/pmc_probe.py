#!/usr/bin/env python3
"""Per-dispatch averages of a few rocprofv3 PMC counters for one kernel of tools/conv_probe.py (counters only, no traces):

    tools/pmc_probe.py <out.json> <kernel-substring> "<counters pass 1>" ["<counters pass 2>" ...] -- <conv_probe args>
"""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out_json, substr = sys.argv[1], sys.argv[2]
    sep = sys.argv.index("--")
    passes, probe = sys.argv[3:sep], sys.argv[sep + 1:]
    res = {}
    for ctrs in passes:
        d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", *ctrs.split(), "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.join(REPO, "tools", "conv_probe.py"), *probe]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, timeout=150, text=True)
        except subprocess.TimeoutExpired:      # some counter combinations never finish on this stack: skip the pass
            res[ctrs] = {"error": "timed out"}
            continue
        acc = {}
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if substr not in row["Kernel_Name"]:
                    continue
                a = acc.setdefault(row["Counter_Name"], [0.0, 0, 0.0])
                a[0] += float(row["Counter_Value"])
                a[1] += 1
                a[2] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
        if not acc:
            res[ctrs] = {"error": r.stdout[-600:]}
        for k, v in acc.items():
            res[k] = round(v[0] / v[1], 1)
            res["avg_us"] = round(v[2] / v[1], 2)
            res["dispatches"] = v[1]
    if "GRBM_GUI_ACTIVE" in res and "SQ_VALU_MFMA_BUSY_CYCLES" in res:
        gui = res["GRBM_GUI_ACTIVE"] / 8.0
        res["derived"] = {"clock_ghz": round(gui / (res["avg_us"] * 1e3), 3),
                          "mfma_busy_frac": round(res["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui), 4)}
    json.dump(res, open(out_json, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
