#!/bin/bash
# Kernel trace of a few guided steps -> tools/gap_probe.py (busy fraction, per-kernel table, split-K combine accounting).
# usage (GPU box): tools/combine_trace.sh [steps]   -> gpurun_out/prof/step_kernel_table.txt
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out/prof; mkdir -p "$out"; rm -rf "$out/ktc"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d "$out/ktc" -- \
    python "$repo/bench.py" --steps ${1:-8} --warmup 2 --cpu-steps 0 --secondary-steps 0 --pmc off > "$out/ktc.log" 2>&1
python "$repo/tools/gap_probe.py" "$out/ktc/**/*kernel_trace.csv" > "$out/step_kernel_table.txt" 2>&1
rm -rf "$out/ktc"
tail -5 "$out/step_kernel_table.txt"
