mkdir -p gpurun_out/r2f; O=$(pwd)/gpurun_out/r2f; R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 4 --warmup 1 --cpu-steps 0 > $O/kt.log 2>&1
f=$(find $O/kt -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats.csv
find $O/kt -name '*kernel_trace.csv' -delete; rm -rf $O/kt
head -40 $O/kernel_stats.csv | cut -c1-220
