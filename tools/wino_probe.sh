cd /root/repo
timeout 900 python -m pytest tests/test_winograd_gpu.py -x -q 2>&1 | tail -15
timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_sampler_gpu.py -x -q 2>&1 | tail -8
for w in 1 0; do OSM_WINOGRAD=$w timeout 600 python bench.py --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('winograd=$w', d['value'], d['ms_per_step'], d.get('kernel_breakdown_ms_per_step'), d['roofline'].get('frac'), d['roofline'].get('kernel'))"; done
