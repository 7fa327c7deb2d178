mkdir -p gpurun_out/r2c; O=gpurun_out/r2c
(timeout 700 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -q -x -k "conv or attn" 2>&1 | tail -25) > $O/t_ops.log
(timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_sampler_gpu.py -q -x -s 2>&1 | grep -v "^\s*$" | tail -25) > $O/t_int.log
SH="--shape 1,8,8,1024,1024,3 --shape 1,16,16,1024,1024,3 --shape 1,32,32,512,512,3 --shape 1,8,8,1024,1024,1 --shape 1,16,16,1024,3072,1 --shape 1,32,32,512,1536,1 --shape 1,8,8,3072,1024,1 --shape 1,16,16,2048,1024,3"
for m in 256 0 1024; do echo "== OSM_SKINNY_MAXM=$m"; OSM_SKINNY_MAXM=$m timeout 300 python tools/conv_probe.py $SH --iters 50 --check; done > $O/probe.txt 2>&1
timeout 400 python bench.py --steps 8 --warmup 2 --cpu-steps 0 --dump-layers $O/layers.json > $O/bench_new.json 2> $O/bench.err
OSM_SKINNY_MAXM=0 OSM_ATTN_FLASH=0 timeout 400 python bench.py --steps 8 --warmup 2 --cpu-steps 0 > $O/bench_base.json 2>> $O/bench.err
OSM_SKINNY_MAXM=1024 timeout 400 python bench.py --steps 8 --warmup 2 --cpu-steps 0 > $O/bench_sk1024.json 2>> $O/bench.err
timeout 400 python bench.py --steps 8 --warmup 2 --cpu-steps 0 --conv-mode f16 --dump-layers $O/layers_f16.json > $O/bench_f16.json 2>> $O/bench.err
tail -3 $O/t_ops.log $O/t_int.log; cat $O/probe.txt
