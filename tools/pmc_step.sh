#!/bin/bash
# usage: tools/pmc_step.sh <out_dir> : SQ counter passes (counters only) over TWO guided steps of bench.py; prints the per-dispatch
# averages of the Winograd kernel by grid size (in-step counterpart of tools/pmc_probe.sh)
out=$1; mkdir -p "$out"; out=$(readlink -f "$out")
repo=$(readlink -f "$(dirname "$0")/..")
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 5 400 rocprofv3 --pmc $grp --output-format csv -d "$out/g$i" -- python $repo/bench.py --steps 2 --warmup 1 --cpu-steps 0 --pmc off --secondary-steps 0 > "$out/g$i.log" 2>&1
done
python - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv3_wino8" not in k: continue
        acc[(r.get("Grid_Size"), r.get("LDS_Block_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -len(list(kv[1].values())[0])):
    print("grid", k, "dispatches", len(list(d.values())[0]))
    for c, v in sorted(d.items()):
        print(f"   {c:34s} avg={sum(v)/len(v):16.1f}")
PY
