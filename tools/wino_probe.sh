cd /root/repo
timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_sampler_gpu.py -x -q 2>&1 | tail -3
for w in 1; do OSM_WINOGRAD=$w timeout 600 python bench.py --steps 6 --warmup 2 --dump-layers gpurun_out/layers_w$w.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('winograd=$w', d['value'], d['ms_per_step'], d.get('kernel_breakdown_ms_per_step'), {k:v for k,v in d['roofline'].items() if k in ('frac','kernel','achieved','executed_tflops','winograd_share_of_time')})"; done
