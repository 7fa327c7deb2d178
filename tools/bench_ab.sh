#!/bin/bash
# usage: tools/bench_ab.sh [rounds] : the whole guided step (bench.py, headline leg only) with the in-tree library and every
# tools/variants/libosm_*.so, round-robin on ONE box; prints ms_per_step and the per-class split of each run
cd "$(dirname "$0")/.."
R=${1:-2}
for r in $(seq $R); do
  for f in normal tools/variants/libosm_*.so; do
    if [ "$f" = normal ]; then unset OSM_LIB; else export OSM_LIB=$(readlink -f $f); fi
    timeout 300 python bench.py --steps 20 --warmup 3 --cpu-steps 0 --pmc off --secondary-steps 0 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_breakdown_ms_per_step']; print('$(basename $f .so)', d['ms_per_step'], ' '.join(f'{a}={b}' for a,b in sorted(k.items())))"
  done
done
