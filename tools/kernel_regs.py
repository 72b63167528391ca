#!/usr/bin/env python3
"""Registers, scratch and size of every kernel in an ISA listing (hipcc -S --cuda-device-only)."""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue

    def g(k):
        r = re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body)
        return r.group(1) if r else "?"

    i = txt.find(name + ":")
    j = txt.find(".end_amdhsa_kernel", i)
    lines = sum(1 for l in txt[i:j].split("\n") if l.startswith("\t") and not l.startswith("\t."))
    print(f"{name[:100]:100s} vgpr {g('next_free_vgpr'):>4s} sgpr {g('next_free_sgpr'):>4s} scratch {g('private_segment_fixed_size'):>5s} insts {lines}")
