cd /root/repo
run() { timeout 900 python bench.py --steps 6 --warmup 2 --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('kernel_breakdown_ms_per_step'))"; }
OSM_LIB=/root/repo/.ab/libosm_prev.so run prev
run head
OSM_LIB=/root/repo/.ab/libosm_prev.so run prev
run head
