#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel of one .hip source (device-only assembly, gfx950): which instances spill.
    python tools/kernel_regs.py osmosis_diffusion_code_amd/csrc/norm.hip [name filter] [-- extra hipcc flags]"""
import re
import subprocess
import sys
import tempfile


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "--" else ""
    extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", f.name] + extra,
                       check=True, stderr=subprocess.DEVNULL)
        txt = open(f.name).read()
    for b in re.split(r"\n  - \.agpr_count:", txt)[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt and flt not in dn:
            continue
        get = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)      # noqa: E731
        agpr = re.match(r"\s*(\d+)", b).group(1)
        print(f"{dn[:90]:90s} vgpr {get('vgpr_count'):>3s} agpr {agpr:>3s} spill {get('vgpr_spill_count'):>3s} "
              f"scratch {get('private_segment_fixed_size'):>5s} lds {get('group_segment_fixed_size'):>6s}")


if __name__ == "__main__":
    main()
