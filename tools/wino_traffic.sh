cd /root/repo
for g in 1 4 8 16; do
  for shape in 1,256,256,256,256,3 1,128,128,512,512,3; do
    OSM_WINO_NGROUP=$g python tools/pmc_probe.py gpurun_out/tr_$g.json conv3_wino "FETCH_SIZE" "WRITE_SIZE" -- --shape $shape --iters 10 --winograd > /dev/null 2>&1
    python - <<EOF
import json
d=json.load(open('gpurun_out/tr_$g.json'))
print('ngroup $g $shape fetch MB', round(d.get('FETCH_SIZE',0)*2*1024/1e6,1), 'write MB', round(d.get('WRITE_SIZE',0)*1024/1e6,1), 'us', d.get('avg_us'))
EOF
  done
done
