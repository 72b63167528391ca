#!/usr/bin/env python3
"""Lints the generated ISA of the kernels that stream postings (place_kernel.hip, team_kernel.hip,
team_stream.hip) for the two hazards hipcc cannot see.

1. The ring's posting loads are issued from inline asm, uncounted by hipcc's s_waitcnt bookkeeping, and waited
   for with hand-counted `s_waitcnt vmcnt(N)`.  A register a load writes is IN FLIGHT from the load until a wait
   that retires it; any instruction -- compiler-made or asm -- that names a register in flight reads stale data
   (or has its result overwritten).  The lint runs the hardware's own rule over the kernel's control-flow graph:
   vector-memory loads return in issue order, so after `s_waitcnt vmcnt(N)` a load is done if at least N loads
   were issued behind it (stores share the counter but may overtake loads: they are not counted as "behind").
   Every load in the listing takes part in the counting, the compiler's as well: a compiler-made load that slipped
   in between the ring's would make a counted wait one too short, and shows here as a ring register read in
   flight.  Only the registers of asm-issued loads are watched (the compiler waits for its own loads itself: a
   spill reload in flight across a call is its business, and the callee begins with a full wait).
   State per program point: {register in flight: loads issued behind its own, at least}; where paths meet, a
   register is in flight if it is on any of them, with the smallest count; iterated to the fixed point (the ring
   loop is cyclic, and its body may hold branches: the two arms of the run-coded refill, the tail that leaves
   padding slots alone).

2. A scalar register written by a vector instruction (v_readlane, v_readfirstlane, a comparison) needs five wait
   states before a vector-memory instruction may read it as its buffer resource or offset.  The ring's refill in
   a stage carries no s_nop of its own (Layout::issue<kSettled>): the stage's other instructions lie between.
   Checked for every buffer load, walking back over EVERY path of the control-flow graph (through the compiler's
   basic-block labels as well as the local labels of an asm statement).

Usage: lint_ring_asm.py place_kernel.s
"""
import re
import sys

LOAD_PREFIXES = ("buffer_load_", "global_load_", "flat_load_", "scratch_load_")
INF = 1 << 20


def vregs_in(text):
    """Every vector register named in an operand string."""
    regs = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        regs |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        regs.add(int(m.group(1)))
    return regs


def regs_of(token):
    token = token.strip().rstrip(",")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", token)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", token)
    return {int(m.group(1))} if m else set()


def sregs_of(token):
    token = token.strip().rstrip(",")
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", token)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", token)
    return {int(m.group(1))} if m else set()


def kernels(lines):
    name, body = None, []
    for line in lines:
        m = re.match(r"^(_ZN8epik_amd\d+(?:place_reads|team_place|team_stream|finish_reads)_kernel\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name:
            body.append(line)
            if line.strip().startswith("s_endpgm"):
                yield name, body
                name = None


class Code:
    """A kernel body as instructions and labels in program order, with its control-flow edges."""

    def __init__(self, body):
        self.ins = []       # (line number in body, text)
        self.in_asm = []    # the instruction stands inside an inline-asm statement
        self.label_at = {}  # label -> index of the instruction behind it
        pending = []
        asm = False
        for n, raw in enumerate(body):
            t = raw.strip()
            if t.startswith(";;#ASMSTART"):
                asm = True
            elif t.startswith(";;#ASMEND"):
                asm = False
            if not t or t.startswith((";", "//")):
                continue
            m = re.match(r"([.\w$]+):", t)
            if m and not t.startswith(("s_", "v_", "ds_", "buffer_", "global_", "flat_", "scratch_")):
                pending.append(m.group(1))
                continue
            if t.startswith("."):
                continue
            for label in pending:
                self.label_at[label] = len(self.ins)
            pending = []
            self.ins.append((n, t))
            self.in_asm.append(asm)
        for label in pending:  # (labels behind the last instruction)
            self.label_at[label] = len(self.ins)
        self.succ = [[] for _ in self.ins]
        for k, (_, t) in enumerate(self.ins):
            w = t.split()
            op = w[0]
            if op == "s_branch":
                if w[1] in self.label_at:
                    self.succ[k].append(self.label_at[w[1]])
                continue
            if op.startswith("s_cbranch") and w[-1] in self.label_at:
                self.succ[k].append(self.label_at[w[-1]])
            if op in ("s_endpgm", "s_setpc_b64"):
                continue
            if k + 1 < len(self.ins):
                self.succ[k].append(k + 1)
        self.pred = [[] for _ in self.ins]
        for k, out in enumerate(self.succ):
            for s in out:
                if s < len(self.ins):
                    self.pred[s].append(k)


def operands_of(text):
    parts = text.split(None, 1)
    return parts[0], (parts[1] if len(parts) > 1 else "")


def is_load(op, rest):
    if op.startswith(LOAD_PREFIXES):
        return True
    # an atomic that returns its old value comes back through the same counter, in load order
    return "_atomic_" in op and (" glc" in rest or " sc0" in rest)


def lint_in_flight(name, body, problems):
    """Hazard 1: no instruction names a register whose load has not been waited for."""
    code = Code(body)
    n = len(code.ins)
    if n == 0:
        return 0
    state_in = [None] * n   # None: not reached yet; else {reg: loads issued behind it}
    state_in[0] = {}
    work = [0]
    reported = {}
    loads_seen = 0

    def step(k, state, report):
        line, t = code.ins[k]
        op, rest = operands_of(t)
        out = state
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", rest)
            if m:
                keep = int(m.group(1))
                out = {r: c for r, c in state.items() if c < keep}
            return out
        named = vregs_in(rest)
        hit = named & state.keys() if state else set()
        if hit and report:
            reported.setdefault(line, f"{name}: line {line}: `{t}` names v{sorted(hit)[0]} while its load is in flight "
                                      f"({state[sorted(hit)[0]]} load(s) issued behind it, no wait has retired it)")
        if op == "s_swappc_b64" and state and report:
            reported.setdefault(line, f"{name}: line {line}: a call with {len(state)} register(s) of the ring in flight")
        if is_load(op, rest):
            dst = regs_of(rest.split(",")[0])
            out = {r: min(c + 1, INF) for r, c in state.items() if r not in dst}
            if code.in_asm[k]:  # (a compiler-made load into a register: the register is the compiler's from here on)
                for r in dst:
                    out[r] = 0
        return out

    while work:
        k = work.pop()
        state = state_in[k]
        out = step(k, state, False)
        for s in code.succ[k]:
            if s >= n:
                continue
            old = state_in[s]
            if old is None:
                new = dict(out)
            else:
                new = dict(old)
                for r, c in out.items():
                    new[r] = min(c, old[r]) if r in old else c
            if new != old:
                state_in[s] = new
                work.append(s)
    for k in range(n):
        if state_in[k] is not None:
            step(k, state_in[k], True)
            op, rest = operands_of(code.ins[k][1])
            if op.startswith("buffer_load_"):
                loads_seen += 1
    problems.extend(reported[line] for line in sorted(reported))
    return loads_seen


def lint_settled(name, body, problems):
    """Hazard 2: five wait states between a vector instruction that writes a scalar register and a buffer load that
    reads it (s_nop N counts N + 1), over every path that leads to the load."""
    code = Code(body)

    def writes_sgpr_from_valu(op):
        return op.startswith(("v_readlane_b32", "v_readfirstlane_b32")) or (op.startswith("v_cmp") and "_e64" in op)

    for k, (line, t) in enumerate(code.ins):
        op, rest = operands_of(t)
        if not op.startswith("buffer_load_"):
            continue
        reads = set()
        for o in rest.split(",")[1:]:
            o = o.strip()
            if o:
                reads |= sregs_of(o.split()[0])
        found = None
        seen = set()
        stack = [(p, 0) for p in code.pred[k]]
        while stack and found is None:
            at, states = stack.pop()
            if states >= 5 or (at, states) in seen:
                continue
            seen.add((at, states))
            u_op, u_rest = operands_of(code.ins[at][1])
            if writes_sgpr_from_valu(u_op):
                dst = sregs_of(u_rest.split(",")[0]) if u_rest else set()
                if dst & reads:
                    found = (sorted(dst & reads)[0], states, code.ins[at][1])
                    break
            cost = int(u_rest) + 1 if u_op == "s_nop" and u_rest.strip().isdigit() else 1
            for p in code.pred[at]:
                stack.append((p, states + cost))
        if found:
            problems.append(f"{name}: line {line}: {op} reads s{found[0]} {found[1]} wait state(s) behind `{found[2]}` (needs 5)")


def lint(path):
    problems = []
    linted = 0
    for name, body in kernels(open(path).read().split("\n")):
        has_ring = any(re.match(r"\s*buffer_load_", l) for l in body)
        if not has_ring:
            continue  # a kernel without a stream (the finish halves of a sharded placement)
        linted += 1
        lint_settled(name, body, problems)
        lint_in_flight(name, body, problems)
    if linted == 0:
        problems.append(f"{path}: no kernel with asm ring loads found")
    return problems


if __name__ == "__main__":
    out = lint(sys.argv[1])
    for p in out[:40]:
        print("LINT:", p)
    print(f"ring-asm lint: {len(out)} problem(s)")
    sys.exit(1 if out else 0)
