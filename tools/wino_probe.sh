cd /root/repo
run() { timeout 600 python bench.py --steps 6 --warmup 2 --cpu-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('kernel_breakdown_ms_per_step'))"; }
cp gpurun_out/libosm_head.so /tmp/libosm_head.so
OSM_LIB=/tmp/libosm_head.so run head_lib
OSM_FUSE_STATS=fwd run new_fwd
OSM_FUSE_STATS=all run new_all
OSM_FUSE_STATS=0 run new_nostats
OSM_LIB=/tmp/libosm_head.so run head_lib_again
