#!/bin/bash
# A/B of the 8-wave Winograd kernel (default) against the 4-wave kernel of round 2 (OSM_WINO8=0): tests, then per-layer times.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_winograd_gpu.py -x -q 2>&1 | tail -5
S="--shape 1,256,256,256,256,3 --shape 1,256,256,512,256,3 --shape 1,128,128,512,512,3 --shape 1,128,128,256,256,3 --shape 1,64,64,512,512,3 --shape 1,64,64,1024,512,3 --shape 1,32,32,512,512,3 --shape 1,16,16,1024,1024,3"
echo "wino4:"; OSM_WINO8=0 timeout 300 python tools/conv_probe.py $S --check --iters 20 --winograd 2>&1 | grep bf16
echo "wino8:"; timeout 300 python tools/conv_probe.py $S --check --iters 20 --winograd 2>&1 | grep bf16
echo "wino8 dgrad:"; timeout 300 python tools/conv_probe.py $S --iters 20 --winograd --dgrad 2>&1 | grep bf16
