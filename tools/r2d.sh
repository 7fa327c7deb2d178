mkdir -p gpurun_out/r2d; O=gpurun_out/r2d
(timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -q -x 2>&1 | tail -40) > $O/t_ops.log
(timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_sampler_gpu.py tests/test_guidance_gpu.py tests/test_prior_sampler.py tests/test_postprocess.py -q -x 2>&1 | tail -40) > $O/t_int.log
B="timeout 400 python bench.py --steps 8 --warmup 2 --cpu-steps 0"
$B --dump-layers $O/layers.json > $O/bench_new.json 2> $O/bench.err
OSM_FUSE_STATS=0 $B > $O/bench_nostats.json 2>> $O/bench.err
OSM_FUSE_STATS=all $B --dump-layers $O/layers_all.json > $O/bench_all.json 2>> $O/bench.err
$B > $O/bench_new2.json 2>> $O/bench.err
OSM_FUSE_STATS=0 $B > $O/bench_nostats2.json 2>> $O/bench.err
$B --conv-mode f16 > $O/bench_f16.json 2>> $O/bench.err
tail -n 4 $O/t_ops.log; tail -n 4 $O/t_int.log
for f in new nostats all new2 nostats2 f16; do python -c "
import json
try:
    d=json.load(open('$O/bench_$f.json'));print('$f',d['value'],d['ms_per_step'],d['roofline']['achieved'],d['kernel_breakdown_ms_per_step'])
except Exception as e: print('$f','ERR',e)"; done
