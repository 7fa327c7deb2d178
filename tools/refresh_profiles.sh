#!/bin/bash
# Regenerates the per-round evidence under gpurun_out/prof/ on a GPU box (copy into profiles/ afterwards):
#   kernel stats (rocprofv3 --kernel-trace --stats), HBM traffic PMC passes (separate runs), layer table, bench line.
# usage: tools/refresh_profiles.sh rNN
set -u
tag=${1:-r01}
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out/prof
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- \
    python "$repo/bench.py" --steps 8 --warmup 1 --cpu-steps 0 --secondary-steps 0 --pmc off > "$out/kt.log" 2>&1
f=$(find "$out/kt" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$out/${tag}_rocprofv3_kernel_stats.csv"
find "$out/kt" -name '*kernel_trace.csv' -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 600 rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -- \
      python "$repo/bench.py" --steps 1 --warmup 1 --cpu-steps 0 --secondary-steps 0 --pmc off > "$out/pmc_$c.log" 2>&1
done
ff=$(find "$out/pmc_FETCH_SIZE" -name '*counter_collection.csv' | head -1)
fw=$(find "$out/pmc_WRITE_SIZE" -name '*counter_collection.csv' | head -1)
python "$repo/tools/pmc_summary.py" "$ff" "$fw" "$out/${tag}_pmc_hbm_traffic.json" > "$out/pmc_summary.txt" 2>&1
rm -rf "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE" "$out/kt"
cd "$repo"
cp "$out/${tag}_pmc_hbm_traffic.json" profiles/ 2>/dev/null   # so that the bench line below carries the new traffic
# BASELINE configurations 3 (B = 8, fp32 storage) and 5 (B = 32, use_fp16): kernel stats + HBM traffic of one step each
for cfg in 3 5; do
  cd /tmp
  timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt$cfg" -- \
      python "$repo/bench.py" --config $cfg --steps 2 --warmup 1 > "$out/kt$cfg.log" 2>&1
  f=$(find "$out/kt$cfg" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$out/${tag}_cfg${cfg}_rocprofv3_kernel_stats.csv"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 600 rocprofv3 --pmc $c --output-format csv -d "$out/pmc${cfg}_$c" -- \
        python "$repo/bench.py" --config $cfg --steps 1 --warmup 1 > "$out/pmc${cfg}_$c.log" 2>&1
  done
  ff=$(find "$out/pmc${cfg}_FETCH_SIZE" -name '*counter_collection.csv' | head -1)
  fw=$(find "$out/pmc${cfg}_WRITE_SIZE" -name '*counter_collection.csv' | head -1)
  python "$repo/tools/pmc_summary.py" "$ff" "$fw" "$out/${tag}_cfg${cfg}_pmc_hbm_traffic.json" > "$out/pmc_summary_cfg$cfg.txt" 2>&1
  rm -rf "$out/pmc${cfg}_FETCH_SIZE" "$out/pmc${cfg}_WRITE_SIZE" "$out/kt$cfg"
  cd "$repo"
done
python bench.py --steps 20 --warmup 3 --dump-layers "$out/${tag}_conv_layer_timings.json" 2> "$out/bench.err" > "$out/${tag}_bench.json"
python "$repo/tools/mfma_busy.py" "$out/${tag}_pmc_mfma_busy.json" > "$out/mfma_busy.txt" 2>&1
tail -c 600 "$out/${tag}_bench.json"; cat "$out/pmc_summary.txt" | head -8
