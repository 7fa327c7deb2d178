cd /root/repo
S="--shape 1,256,256,256,256,3 --shape 1,128,128,512,512,3 --shape 1,128,128,256,256,3 --shape 1,64,64,512,512,3 --shape 1,64,64,1024,512,3 --shape 1,32,32,512,512,3 --shape 1,16,16,1024,1024,3"
timeout 300 python tools/conv_probe.py $S --check --iters 20 --winograd 2>&1 | grep wino
