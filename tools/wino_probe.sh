cd /root/repo
timeout 900 python -m pytest tests/test_winograd_gpu.py -x -q 2>&1 | tail -8
timeout 900 python -m pytest tests/test_fp16_gpu.py -x -q 2>&1 | tail -5
run() { timeout 900 python bench.py --steps 4 --warmup 1 --cpu-steps 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d.get('kernel_breakdown_ms_per_step'), d['roofline']['frac'])"; }
run --conv-mode f16
OSM_WINOGRAD=0 run --conv-mode f16
run --batch 8 --conv-mode f16
