#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE collected in SEPARATE runs, as
MI355X_MICROARCH.md section HBM prescribes) per kernel.

    tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]

Units / corrections (guide): FETCH_SIZE and WRITE_SIZE are in KiB of memory-side (fabric) requests;
on gfx950 FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced streaming
reads, so the read side is doubled for kernels whose loads are 16 B/lane (all of ours are).
WRITE_SIZE is used as reported (uncalibrated per the guide)."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    return acc


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in sorted(fetch, key=lambda k: -fetch[k][0]):
        f, n, us = fetch[k]
        w = write.get(k, [0.0, 0, 0.0])
        rd = 2.0 * f * 1024 / n          # gfx950 correction: x2 on 16 B/lane streaming reads
        wr = (w[0] * 1024 / w[1]) if w[1] else 0.0
        rows.append({"kernel": k, "launches": n, "avg_us_under_pmc": round(us / n, 2),
                     "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                     "hbm_bytes_per_launch": round(rd + wr)})
    out = {"note": "read = 2 x FETCH_SIZE KiB (gfx950 half-count correction), write = WRITE_SIZE KiB; per launch averages",
           "kernels": rows}
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt)
    for r in rows[:14]:
        print(f'{r["kernel"]:42s} n={r["launches"]:5d}  read {r["read_bytes_per_launch"]/1e6:9.2f} MB  '
              f'write {r["write_bytes_per_launch"]/1e6:9.2f} MB')


if __name__ == "__main__":
    main()
