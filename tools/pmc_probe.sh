#!/bin/bash
# usage: tools/pmc_probe.sh <out_dir> <probe args...>   -- several rocprofv3 --pmc passes over tools/conv_probe.py
# (counters only: never combined with sys/hip/hsa traces)
out=$(readlink -f "$1" 2>/dev/null || echo "$1"); shift
mkdir -p "$out"; out=$(readlink -f "$out")
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $grp --output-format csv -d "$out/g$i" -- python /root/repo/tools/conv_probe.py "$@" > "$out/g$i.log" 2>&1
done
python - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pack_weight" in k or "splitk_reduce" in k or "Cijk" in k or "at::" in k or "rocclr" in k: continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")
PY
