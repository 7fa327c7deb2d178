#!/bin/bash
# Samples rocm-smi (power, clocks, performance level) while the dominant convolution runs back to back:
# evidence for the clock the power manager grants under the bf16x6 MFMA stream.   usage: tools/power_probe.sh <out.txt> [extra conv_probe args, e.g. --winograd]
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$1; shift
shape=${SHAPE:-1,256,256,256,256,3}      # SHAPE=32,256,256,256,256,3 with --mode f16: the dominant kernel of BASELINE config 5
iters=${ITERS:-60000}
{
  echo "== idle"; rocm-smi --showpower --showclocks --showperflevel --showmaxpower 2>&1 | grep -v "^$" | head -40
  python "$repo/tools/conv_probe.py" --shape $shape --iters $iters "$@" > /tmp/probe_busy.txt 2>&1 &
  pid=$!
  for w in $(seq 1 170); do     # wait for the load (the first import of torch on a fresh box takes 1-2 minutes)
    pw=$(rocm-smi --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$" | head -1)
    [ -n "$pw" ] && [ "${pw%.*}" -gt 450 ] && break
    sleep 1
  done
  for i in 1 2 3 4 5; do echo "== under load, sample $i"; rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk|socclk" ; sleep 1; done
  kill $pid 2>/dev/null; wait $pid 2>/dev/null
} > "$out" 2>&1
