#!/bin/bash
# usage: tools/run_variants.sh "<conv_probe args>" : the normal build, then every tools/variants/libosm_*.so
cd "$(dirname "$0")/.."
echo "== normal"; timeout 200 python tools/conv_probe.py $1 2>&1 | grep -v "^$"
for f in tools/variants/libosm_*.so; do
  echo "== $(basename $f .so)"
  OSM_LIB=$(readlink -f $f) timeout 200 python tools/conv_probe.py $1 2>&1 | grep -v "^$"
done
