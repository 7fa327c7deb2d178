cd /root/repo
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fp16_gpu.py tests/test_optin_kernels_gpu.py -x -q 2>&1 | tail -3
S="--shape 1,256,256,256,256,3 --shape 1,128,128,512,512,3 --shape 1,8,8,1024,1024,3 --shape 1,256,256,4,256,3"
timeout 300 python tools/conv_probe.py $S --check --iters 20 2>&1 | grep bf16
timeout 300 python tools/conv_probe.py --mode f16 $S --check --iters 20 2>&1 | grep f16
run() { timeout 900 python bench.py --steps 4 --warmup 1 --cpu-steps 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d.get('kernel_breakdown_ms_per_step'), d['roofline']['frac'])"; }
run
run --conv-mode f16
run --batch 8 --conv-mode f16
