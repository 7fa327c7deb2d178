mkdir -p gpurun_out/r2e; O=gpurun_out/r2e
(OSM_TALL_MINM=1 timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -q -x -k "conv" 2>&1 | tail -5) > $O/t_tall.log
(timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "attn_flash or column" 2>&1 | tail -5) > $O/t_flash.log
SH="--shape 1,256,256,256,256,3 --shape 1,256,256,512,256,3 --shape 1,128,128,512,512,3 --shape 1,128,128,256,256,3 --shape 1,64,64,512,512,3 --shape 8,64,64,512,512,3 --shape 8,32,32,512,512,3"
for m in 0 4096; do echo "== OSM_TALL_MINM=$m"; OSM_TALL_MINM=$m timeout 300 python tools/conv_probe.py $SH --iters 30 --check; done > $O/probe.txt 2>&1
for m in 0 4096; do echo "== f16 OSM_TALL_MINM=$m"; OSM_TALL_MINM=$m timeout 300 python tools/conv_probe.py $SH --iters 30 --check --mode f16; done >> $O/probe.txt 2>&1
B="timeout 400 python bench.py --steps 8 --warmup 2 --cpu-steps 0"
$B > $O/bench_base.json 2> $O/bench.err
OSM_TALL_MINM=16384 $B > $O/bench_tall16k.json 2>> $O/bench.err
OSM_TALL_MINM=4096 $B > $O/bench_tall4k.json 2>> $O/bench.err
OSM_TALL_MINM=16384 $B --conv-mode f16 > $O/bench_f16_tall.json 2>> $O/bench.err
tail -n 3 $O/t_tall.log; tail -n 3 $O/t_flash.log; cat $O/probe.txt | grep -v amdgpu.ids
for f in base tall16k tall4k f16_tall; do python -c "
import json
try:
    d=json.load(open('$O/bench_$f.json'));print('$f',d['value'],d['ms_per_step'],d['roofline']['achieved'],d['kernel_breakdown_ms_per_step'])
except Exception as e: print('$f','ERR',e)"; done
