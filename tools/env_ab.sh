#!/bin/bash
cd /root/repo
run() { echo "== $1"; env $1 timeout 300 python bench.py --steps 60 --warmup 5 --cpu-steps 0 --pmc off --secondary-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['finite_outputs'])"; }
for r in 1 2; do
run "OSM_NOP=1"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 HIP_FORCE_DEV_KERNARG=1"
run "GPU_MAX_HW_QUEUES=1"
run "OSM_GRAPH=0"
done
