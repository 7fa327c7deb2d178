#!/bin/bash
# Per-kernel time (conv kernel vs split-K reduce) of the low-resolution conv shapes: rocprofv3 --kernel-trace --stats
# over tools/conv_probe.py.   usage: tools/lowres_probe.sh <out_dir>
repo=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$1"; out=$(cd "$1" && pwd)
cd /tmp && export TMPDIR=/tmp
for s in ${SHAPES:-1,32,32,512,512,3 1,16,16,1024,1024,3 1,8,8,1024,1024,3 1,32,32,512,512,1 1,8,8,1024,1024,1}; do
  tag=$(echo $s | tr , _)
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag" -- \
      python "$repo/tools/conv_probe.py" --shape $s --iters 50 ${PROBE_ARGS:-} > "$out/$tag.log" 2>&1
  f=$(find "$out/$tag" -name '*kernel_stats.csv' | head -1)
  echo "== $s"; [ -n "$f" ] && head -6 "$f" | cut -c1-200
  find "$out/$tag" -name '*kernel_trace.csv' -delete
done
