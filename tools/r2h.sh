mkdir -p gpurun_out/r2h; O=gpurun_out/r2h


B="timeout 400 python bench.py --steps 8 --warmup 2 --cpu-steps 0"
$B > $O/bench_new.json 2> $O/bench.err
$B --batch 8 > $O/bench_b8.json 2>> $O/bench.err
$B --conv-mode f16 --batch 8 > $O/bench_f16_b8.json 2>> $O/bench.err
$B --conv-mode f16 --batch 32 --steps 4 --warmup 1 > $O/bench_f16_b32.json 2>> $O/bench.err

for f in new b8 f16_b8 f16_b32; do python -c "
import json
try:
    d=json.load(open('$O/bench_$f.json'));print('$f',d['value'],d['ms_per_step'],d['roofline']['achieved'],d['kernel_breakdown_ms_per_step'])
except Exception as e: print('$f','ERR',e)"; done
