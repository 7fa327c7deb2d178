#!/bin/bash
# usage: tools/env_ab.sh [rounds] "ENV=V ..." "ENV=V ..." ... : the whole guided step (bench.py, headline leg only, 60 steps) in separate processes,
# one per environment setting, round-robin on ONE box -- for switches the library reads once per process
cd "$(dirname "$0")/.."
R=${1:-2}; shift
[ $# -eq 0 ] && set -- "OSM_NOP=1" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "GPU_MAX_HW_QUEUES=1" "OSM_GRAPH=0"
for r in $(seq $R); do
  for e in "$@"; do
    echo -n "$e  "
    env $e timeout 300 python bench.py --steps 60 --warmup 5 --cpu-steps 0 --pmc off --secondary-steps 0 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'attention', d['kernel_breakdown_ms_per_step'].get('attention_core'), d['config']['finite_outputs'])"
  done
done
