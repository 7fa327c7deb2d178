#!/usr/bin/env python3
"""Kernel-busy time vs wall time over the timed steps of a bench.py run, from a rocprofv3 --kernel-trace CSV.

    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- \
        python bench.py --steps 20 --warmup 3 --cpu-steps 0 --secondary-steps 0
    tools/gap_probe.py $OUT/**/*kernel_trace.csv
The window is the span between the first and the last `phys_finalize_kernel` (only the sampler's steps launch it); prints
wall, busy (union of kernel intervals), the gap histogram and the kernels that follow the longest gaps.
"""
import csv
import glob
import sys
from collections import Counter


def short_name(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n[:n.index("(")] if "(" in n else n


def main():
    files = [f for a in sys.argv[1:] for f in glob.glob(a, recursive=True)]
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if "phys_finalize_kernel" in r[2]]
    if not idx:
        raise SystemExit("no phys_finalize_kernel in the trace")
    # steps are separated by > 1 UNet's worth of kernels between finalize launches; skip the first quarter (warm-up)
    lo, hi = idx[len(idx) // 4], idx[-1]
    win = rows[lo:hi + 1]
    wall = win[-1][1] - win[0][0]
    busy, end = 0, win[0][0]
    gaps = []
    for k, (s, e, n) in enumerate(win):
        if s > end:
            gaps.append((s - end, n, win[k - 1][2] if k else ""))
            busy += e - s
        elif e > end:
            busy += e - end
        end = max(end, e)
    nfin = sum(1 for r in win if "phys_finalize_kernel" in r[2])
    print(f"window: {len(win)} kernels, {nfin} phys_finalize launches, wall {wall / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms "
          f"({busy / wall:.1%}), idle {(wall - busy) / 1e6:.2f} ms in {len(gaps)} gaps")
    per = Counter()
    cnt = Counter()
    for s0, e0, n in win:
        k = n.replace("void ", "").replace("(anonymous namespace)::", "")
        k = k[:k.index("(")] if "(" in k else k
        per[k] += e0 - s0
        cnt[k] += 1
    npost = sum(1 for r in win if "posterior_kernel" in r[2])      # one per guided step
    steps = npost if npost else max(1, round(nfin / 21))
    print(f"per step (~{steps} steps in the window): kernel time by kernel")
    for k, v in per.most_common(40):
        print(f"  {v / 1e6 / steps:7.3f} ms  {cnt[k] / steps:7.1f} launches  {v / cnt[k] / 1e3:7.1f} us  {k}")
    h = Counter()
    for g, _, _ in gaps:
        h[min(int(g / 1000), 20)] += g
    print("idle time by gap length (us: total ms):", {k: round(v / 1e6, 2) for k, v in sorted(h.items())})
    after = Counter()
    for g, n, p in gaps:
        after[(p[:50], n[:50])] += g
    for (p, n), v in after.most_common(15):
        print(f"  {v / 1e6:7.2f} ms idle between  {p}  ->  {n}")
    # ---- split-K combines (VERDICT r04 item 4 iii): is a combine hidden behind its producer?  For every splitk_reduce* launch: its
    # duration, the idle gap in front of it (producer end -> combine start) and how much of it ran while the producer was still running
    tot = {"n": 0, "dur": 0, "gap": 0, "overlap": 0, "after": 0}
    by = Counter()
    for k in range(1, len(win) - 1):
        s0, e0, n = win[k]
        if "splitk_reduce" not in n:
            continue
        ps, pe, pn = win[k - 1]
        ns = win[k + 1][0]
        tot["n"] += 1
        tot["dur"] += e0 - s0
        tot["gap"] += max(0, s0 - pe)
        tot["overlap"] += max(0, min(e0, pe) - s0)
        tot["after"] += max(0, ns - e0)
        short = pn.replace("void ", "").replace("(anonymous namespace)::", "")
        by[short[:short.index("(")] if "(" in short else short] += 1
    if tot["n"]:
        print(f"split-K combines per step: {tot['n'] / steps:.1f} launches, kernel time {tot['dur'] / 1e6 / steps:.3f} ms, idle gap producer -> "
              f"combine {tot['gap'] / 1e6 / steps:.3f} ms, time overlapped with the producer {tot['overlap'] / 1e6 / steps:.3f} ms, idle gap combine -> "
              f"next kernel {tot['after'] / 1e6 / steps:.3f} ms")
        print("  producers:", dict(by.most_common(6)))
        # what consumes a combined tensor next (a GroupNorm that follows could do the combine in its own load stage)
        nxt = Counter()
        for k in range(1, len(win) - 1):
            if "splitk_reduce" in win[k][2]:
                nn = win[k + 1][2].replace("void ", "").replace("(anonymous namespace)::", "")
                nxt[(short_name(win[k][2]), nn[:nn.index("(")] if "(" in nn else nn)] += 1
        for (a, b), v in nxt.most_common(30):
            print(f"  {v / steps:6.1f} per step  {a}  ->  {b}")


if __name__ == "__main__":
    main()
