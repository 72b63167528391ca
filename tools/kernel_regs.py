#!/usr/bin/env python3
"""Registers, spills and scratch of every kernel in a hipcc -S listing (make -C epik_amd/csrc asm)."""
import re
import subprocess
import sys


def main(paths):
    for f in paths:
        txt = open(f).read()
        for b in txt.split("  - .agpr_count:")[1:]:
            nm = re.search(r"\.name:\s+(\S+)", b).group(1)
            vg = re.search(r"\.vgpr_count:\s+(\d+)", b).group(1)
            sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", b).group(1)
            sc = re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1)
            dn = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
            dn = dn.replace("epik_amd::", "").replace("(anonymous namespace)::", "")
            print(f"{vg:>4} vgpr {sp:>3} spill {sc:>5} scratch  {dn[:120]}")


if __name__ == "__main__":
    main(sys.argv[1:] or ["gpurun_out/team_stream.s", "gpurun_out/team_kernel.s", "gpurun_out/place_kernel.s"])
