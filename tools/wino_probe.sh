#!/bin/bash
# Quick check of the Winograd kernel on a GPU box: its tests, the per-layer A/B against the direct kernel, one bench line.
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_winograd_gpu.py -x -q 2>&1 | tail -3
S="--shape 1,256,256,256,256,3 --shape 1,128,128,512,512,3 --shape 1,128,128,256,256,3 --shape 1,64,64,512,512,3 --shape 1,64,64,1024,512,3 --shape 1,32,32,512,512,3 --shape 1,16,16,1024,1024,3"
echo "direct:"; timeout 300 python tools/conv_probe.py $S --check --iters 20 2>&1 | grep bf16
echo "winograd:"; timeout 300 python tools/conv_probe.py $S --check --iters 20 --winograd 2>&1 | grep bf16
timeout 600 python bench.py --steps 6 --warmup 2 --cpu-steps 0 2>/dev/null | tail -1 | cut -c1-400
