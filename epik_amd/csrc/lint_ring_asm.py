#!/usr/bin/env python3
"""Lints the generated ISA of the kernels that stream postings (place_kernel.hip, team_kernel.hip,
team_stream.hip) for the one hazard hipcc cannot see.

The ring's posting loads are issued from inline asm (uncounted by hipcc's s_waitcnt
bookkeeping), so a compiler-generated copy of a ring register made while its load is
still in flight would capture stale data.  Every ring stage is, in program order,

    asm: s_waitcnt vmcnt(N) ; v_mad / v_mov    wait for slot i and move its values out
    compiler code                              LDS read-add-write on the moved-out values
    asm: buffer_load_* -> slot i               refill of the slot

so a ring register is in flight from the asm load that writes it until an asm VALU
instruction reads it behind a wait; a compiler-generated instruction that names a register
in flight fails the lint.
Usage: lint_ring_asm.py place_kernel.s
"""
import re
import sys


def regs_of(token):
    token = token.strip().rstrip(",")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", token)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", token)
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


def lint_loop(name, body, events, problems):
    """One ring loop: `events` are its asm statements in program order."""
    # The ring of a round that starts empty is filled by a first trip of loads alone (no stage, no wait): straight-line
    # code in front of the loop, in which a slot's registers are in flight from its load on and free before it.
    k = 0
    while k < len(events) and events[k][1] == "load":
        k += 1
    if 2 <= k < len(events):
        start = events[0][0]
        while start > 0 and not body[start].strip().startswith(";;#ASMSTART"):
            start -= 1
        _walk(name, body, start, events[k][0] - 1, set(), problems)
        events = events[k:]
    ring = set()
    for e in events:
        if e[1] == "load":
            ring |= e[2]
    waits = [e[0] for e in events if e[1] == "wait"]
    loads = [e[0] for e in events if e[1] == "load"]
    if not loads or not waits:
        return
    # From the loop's first asm statement to a margin behind its last asm load (the
    # fall-through into the tail, which still runs before the drain).  Walk it in program
    # order with a per-register state: a slot register is IN FLIGHT from its asm load until an
    # asm vector instruction reads it (behind the stage's asm wait); from there to the refill
    # the compiler may reuse it.  The loop is cyclic, so at its top every slot counts as in flight.
    lo, hi = max(0, min(waits[0], loads[0]) - 1), loads[-1] + 15  # -1: the ;;#ASMSTART line
    for back in range(lo, max(lo - 40, -1), -1):  # ... from the loop's label on (the pipelined ring reads LDS before its first asm)
        if body[back].startswith(".LBB"):
            lo = back
            break
    # Which slots are in flight where the loop is entered again: those its last trip left in flight behind
    # its last asm load -- all of them for the plain ring, all but the cell of slot 0 for the pipelined one
    # (its last stage has already waited for slot 0 and turned its cell into addresses).  One silent walk
    # from "everything in flight" finds that state; the walk that reports starts from it.
    end = loads[-1] + 1
    if body[lo].startswith(".LBB"):  # ... to the branch back to the loop's label
        label = body[lo].split(":")[0]
        for n in range(loads[-1], min(loads[-1] + 60, len(body))):
            t = body[n].strip()
            if t.startswith(("s_cbranch", "s_branch")) and t.split()[-1] == label:
                end = n
                break
    inflight = _walk(name, body, lo, end, set(ring), None)
    _walk(name, body, lo, hi, inflight, problems)


def _walk(name, body, lo, hi, inflight, problems):
    inflight = set(inflight)
    in_asm = False
    skip_to = None  # behind an unconditional branch forwards: the walk goes on at its target (what lies between is
                    # reached from elsewhere -- the path that skipped the loop, on which nothing is in flight)
    for n in range(lo, min(hi, len(body) - 1) + 1):
        s = body[n].strip()
        if skip_to is not None:
            if s.startswith(skip_to + ":"):
                skip_to = None
            continue
        if not in_asm and s.startswith("s_branch "):
            target = s.split()[1]
            if any(body[m].strip().startswith(target + ":") for m in range(n + 1, min(n + 80, len(body)))):
                skip_to = target  # (the target may lie behind `hi`: then the walk ends inside the skipped block)
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith((";", ".")):
            continue
        parts = s.split(None, 1)
        operands = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if in_asm:
            if s.startswith(("buffer_load_", "global_load_")):
                inflight |= regs_of(operands[0])
            elif s.startswith("v_") and len(operands) >= 2:
                # consumed (v_mov / v_mad) behind the wait of this same asm statement
                for o in operands[1:]:
                    inflight -= regs_of(o)
            continue
        touched = set()
        for o in operands:
            if o:
                touched |= regs_of(o.split()[0])
        # `v_mov vX, <constant>` is how hipcc initialises the empty ring on the path that skips
        # the loop (laid out behind it): it reads no register, and nothing is in flight there
        if parts[0].startswith("v_mov_b32") and len(operands) == 2 and not regs_of(operands[1]) \
                and not operands[1].startswith(("s", "v")):
            continue
        if touched & inflight and problems is not None:
            problems.append(f"{name}: line {n}: compiler code touches an in-flight ring register: {s}")
    return inflight


def sregs_of(token):
    token = token.strip().rstrip(",")
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", token)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", token)
    return {int(m.group(1))} if m else set()


def lint_settled(name, body, problems):
    """A scalar register written by a vector instruction (v_readlane, v_readfirstlane, a comparison) needs five
    wait states before a vector-memory instruction may read it as its buffer resource or offset.  The ring's
    refill in a stage carries no s_nop of its own (Layout::issue<kSettled>): the stage's other instructions lie
    between.  Checked here for every buffer load: walking back over every path through straight-line code and
    the local labels of an asm statement (the two arms of the run-coded refill), no such write to one of its
    scalar operands within the last five wait states (s_nop N counts N + 1).  A basic-block label of the
    compiler's ends the walk: the first instructions behind one are not checked against what precedes them on
    other paths -- which is why the ring's first trip, straight from the lanes, keeps its s_nop 4."""
    code = []  # the instructions in program order: (line number, text); labels as (line number, "LABEL", name)
    for n, l in enumerate(body):
        t = l.strip()
        if not t or t.startswith((";", "//")):
            continue
        m = re.match(r"(\.L\w+):", t)
        if m:
            code.append((n, "LABEL", m.group(1)))
            continue
        if t.startswith("."):
            continue
        code.append((n, t, None))

    def walk(back, states, reads, load_line, load_text):
        while back >= 0 and states < 5:
            n, u, label = code[back]
            if u == "LABEL":
                if label.startswith(".LBB"):
                    return
                # a label inside an asm statement: from the branches to it, and from above unless an
                # unconditional branch stands there
                for src in range(back - 1, max(back - 40, -1), -1):
                    w = code[src][1].split()
                    if w and w[0].startswith(("s_cbranch", "s_branch")) and w[-1] == label:
                        walk(src, states, reads, load_line, load_text)
                if back > 0 and code[back - 1][1].startswith("s_branch "):
                    return
                back -= 1
                continue
            w = u.split(None, 1)
            if w[0].startswith(("v_readlane_b32", "v_readfirstlane_b32")) or (w[0].startswith("v_cmp") and "_e64" in w[0]):
                dst = sregs_of(w[1].split(",")[0]) if len(w) > 1 else set()
                if dst & reads:
                    problems.append(f"{name}: line {load_line}: {load_text.split()[0]} reads s{sorted(dst & reads)[0]} "
                                    f"{states} wait state(s) behind `{u}` (needs 5)")
                    return
            states += int(w[1]) + 1 if w[0] == "s_nop" and len(w) > 1 and w[1].strip().isdigit() else 1
            back -= 1

    for k, (n, t, _) in enumerate(code):
        if not t.startswith("buffer_load_"):
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        reads = set()
        for o in ops[1:]:
            reads |= sregs_of(o.split()[0])
        before = len(problems)
        walk(k - 1, 0, reads, n, t)
        del problems[before + 1:]  # (one report per load)


def lint(path):
    problems = []
    linted = 0
    for name, body in kernels(open(path).read().split("\n")):
        lint_settled(name, body, problems)
        # asm statements in program order: (line, kind, regs), kind in {"wait", "drain", "load"}
        events, in_asm = [], False
        for n, l in enumerate(body):
            s = l.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
            elif s.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm and s.startswith(("buffer_load_", "global_load_")):
                events.append((n, "load", regs_of(s.split()[1])))
            elif in_asm and s.startswith("s_waitcnt vmcnt("):
                events.append((n, "drain" if "vmcnt(0)" in s else "wait", set()))
        if not any(e[1] == "load" for e in events):
            continue  # a kernel without a stream (the finish halves of a sharded placement)
        linted += 1
        # the kernel may hold several copies of the ring loop (first pass / further passes):
        # each ends with its tail's `s_waitcnt vmcnt(0)`
        group = []
        for e in events:
            if e[1] == "drain":
                lint_loop(name, body, group, problems)
                group = []
            else:
                group.append(e)
        lint_loop(name, body, group, problems)
    if linted == 0:
        problems.append(f"{path}: no kernel with asm ring loads found")
    return problems


if __name__ == "__main__":
    out = lint(sys.argv[1])
    for p in out[:40]:
        print("LINT:", p)
    print(f"ring-asm lint: {len(out)} problem(s)")
    sys.exit(1 if out else 0)
