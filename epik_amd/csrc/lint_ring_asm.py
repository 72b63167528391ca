#!/usr/bin/env python3
"""Lints the generated ISA of place_kernel.hip for the one hazard hipcc cannot see:
the ring's posting loads are issued from inline asm (uncounted by hipcc's s_waitcnt
bookkeeping), so any compiler-generated instruction that READS or WRITES a ring
register outside the consume blocks could touch it while its load is in flight.

Rule checked per kernel: every register that is the destination of an asm
`buffer_load_dwordx2` may appear, outside ;;#ASMSTART/;;#ASMEND blocks, only
 (a) as a source of the consume instructions (v_lshlrev_b32 / ds_add_f32 / v_add_f32
     operands that follow an asm `s_waitcnt vmcnt(N)`), or
 (b) as the destination of a ds_read_b64 / v_mov that is followed, before the next
     asm load into the same register, by no asm wait (descriptor temporaries and the
     zero-initialisation ahead of the loop).
Anything else -- in particular a v_mov FROM a ring register -- fails the lint.
"""
import re
import sys


def regs_of(token):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", token)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", token)
    return {int(m.group(1))} if m else set()


def lint(path):
    text = open(path).read().split("\n")
    problems = []
    kernel = None
    ring = set()
    body = []
    for line in text:
        m = re.match(r"^(_ZN8epik_amd18place_reads_kernel\w+):", line)
        if m:
            kernel, ring, body = m.group(1), set(), []
            continue
        if kernel and line.startswith("\t.end_amdhsa_kernel"):
            kernel = None
        if kernel:
            body.append(line)
            if line.strip().startswith("s_endpgm"):
                in_asm = False
                for l in body:
                    s = l.strip()
                    if s.startswith(";;#ASMSTART"):
                        in_asm = True
                    elif s.startswith(";;#ASMEND"):
                        in_asm = False
                    elif in_asm and s.startswith("buffer_load_dwordx2"):
                        ring |= regs_of(s.split()[1].rstrip(","))
                # the ring loop is contiguous in the .s: from the first asm load to a margin
                # behind the last one (its exit blocks, which run before the vmcnt(0) drain)
                load_lines = [n for n, l in enumerate(body) if l.strip().startswith("buffer_load_dwordx2")
                              and n > 0 and "s_nop" in body[n - 1]]
                lo, hi = (load_lines[0], load_lines[-1] + 60) if load_lines else (0, -1)
                in_asm = False
                for n, l in enumerate(body):
                    s = l.strip()
                    if s.startswith(";;#ASMSTART"):
                        in_asm = True
                        continue
                    if s.startswith(";;#ASMEND"):
                        in_asm = False
                        continue
                    if in_asm or not s or s.startswith((";", ".")) or not (lo <= n <= hi):
                        continue
                    ops = s.split(None, 1)
                    if len(ops) < 2:
                        continue
                    operands = [o.strip() for o in ops[1].split(",")]
                    srcs = set()
                    for o in operands[1:]:
                        srcs |= regs_of(o.split()[0]) if o else set()
                    if ops[0].startswith("v_mov") and srcs & ring:
                        problems.append(f"{kernel}: copy FROM ring register: {s}")
                body = []
    return problems


if __name__ == "__main__":
    out = lint(sys.argv[1])
    for p in out:
        print("LINT:", p)
    print(f"ring-asm lint: {len(out)} problem(s)")
    sys.exit(1 if out else 0)
