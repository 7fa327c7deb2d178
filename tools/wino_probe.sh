cd /root/repo
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py -x -q 2>&1 | tail -3
S="--shape 1,256,256,256,8,3 --shape 1,256,256,256,4,3 --shape 1,256,256,8,256,3 --shape 1,256,256,4,256,3"
OSM_NARROW=0 timeout 300 python tools/conv_probe.py $S --check --iters 20 2>&1 | grep bf16
timeout 300 python tools/conv_probe.py $S --check --iters 20 2>&1 | grep bf16
timeout 300 python tools/conv_probe.py --mode f16 $S --check --iters 20 2>&1 | grep f16
