#!/bin/bash
# Kernel-level durations (rocprofv3 kernel trace) of the GroupNorm kernels in tools/gn_probe.py: the probe's HIP-event numbers
# include the host's enqueue time (one Python call per launch), which hides anything under ~12 us.
# usage: tools/gn_kernel_times.sh "<gn_probe args>"   -> gpurun_out/gnk/kernel_stats.csv + a per-kernel table on stdout
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out/gnk
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python "$repo/tools/gn_probe.py" $1 > "$out/probe.log" 2>&1
f=$(find "$out/kt" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$out/kernel_stats.csv"
t=$(find "$out/kt" -name '*kernel_trace.csv' | head -1)
python - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"][:70], r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, [])
    a.append(d)
for k, v in agg.items():
    if "gn_" in k[0]:
        v2 = sorted(v)[len(v) // 10: len(v) - len(v) // 10] or v
        print(f"{k[0]:70s} grid {k[1]:>8s} wg {k[2]:>4s} n {len(v):4d}  median {sorted(v)[len(v)//2]:8.1f} us  trimmed mean {sum(v2)/len(v2):8.1f} us")
PY
rm -rf "$out/kt"
