#!/usr/bin/env python3
"""A/B of library builds on one GPU box: runs tools/conv_probe.py for every build round-robin (so that thermal / clock
drift hits all builds alike) and prints the median time per shape.

    tools/ab.py [--rounds 5] [--env NAME=V ...] -- <conv_probe args>
builds = the in-tree library ("normal") + every tools/variants/libosm_*.so; --env adds a build = normal + that environment.
"""
import glob
import os
import re
import statistics
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    sep = sys.argv.index("--")
    opts, probe = sys.argv[1:sep], sys.argv[sep + 1:]
    rounds, envs = 5, []
    i = 0
    while i < len(opts):
        if opts[i] == "--rounds":
            rounds = int(opts[i + 1]); i += 2
        elif opts[i] == "--env":
            envs.append(opts[i + 1]); i += 2
        else:
            raise SystemExit("unknown option " + opts[i])
    builds = [("normal", {})]
    for e in envs:
        k, v = e.split("=", 1)
        builds.append((e, {k: v}))
    for f in sorted(glob.glob(os.path.join(REPO, "tools", "variants", "libosm_*.so"))):
        builds.append((os.path.basename(f)[7:-3], {"OSM_LIB": f}))
    res = {}
    for r in range(rounds):
        for name, env in builds:
            out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "conv_probe.py"), *probe],
                                 env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
            for line in out.splitlines():
                m = re.search(r"(\S+)\s+(\d+,\d+,\d+,\d+,\d+,\d+)\s+splitk=\s*(\d+)\s+([\d.]+) us(.*relerr (\S+))?", line)
                if m:
                    res.setdefault((m.group(2), name), []).append((float(m.group(4)), m.group(6)))
    shapes = sorted({k[0] for k in res}, key=lambda s: [-int(v) for v in s.split(",")])
    for s in shapes:
        print(s)
        base = statistics.median(t for t, _ in res[(s, "normal")])
        for name, _ in builds:
            ts = [t for t, _ in res.get((s, name), [])]
            if not ts:
                print(f"   {name:28s} (no result)")
                continue
            md = statistics.median(ts)
            err = res[(s, name)][0][1]
            print(f"   {name:28s} median {md:8.1f} us  min {min(ts):8.1f}  ({md / base - 1:+.1%})" + (f"  relerr {err}" if err else ""))


if __name__ == "__main__":
    main()
