"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/epik_amd.h declares, validates its arguments, and refuses to compute
without a GPU (there is no CPU fallback in the product path)."""
import ctypes
import os
import re

import numpy as np
import pytest

from epik_amd import alphabet, capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "epik_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(epik_amd_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    declared = _header_symbols()
    assert declared, "no prototypes found in include/epik_amd.h"
    assert sorted(capi.EXPORTS) == declared
    for name in declared:
        assert hasattr(lib, name), f"libepik_amd.so does not export {name}"


def test_struct_layouts_match_header():
    assert capi.PLACEMENT.itemsize == 16
    assert capi.PKDB_VALUE.itemsize == 8
    assert ctypes.sizeof(capi.PlacerDesc) == 104
    assert capi.PlacerDesc.keep_factor.offset == 24
    assert capi.PlacerDesc.offsets.offset == 56
    assert capi.PlacerDesc.device.offset == 80
    assert capi.PlacerDesc.keys.offset == 88 and capi.PlacerDesc.num_present.offset == 96


def _desc(db, **over):
    off = np.ascontiguousarray(db.offsets, dtype=np.uint32)
    cls = alphabet.char_class_table(db.states)
    d = dict(abi_version=capi.ABI_VERSION, kmer_size=db.kmer_size, alphabet_size=db.alphabet_size,
             num_branches=db.num_branches, keep_at_most=7, offset_bits=32, keep_factor=0.01,
             threshold=float(db.threshold), log_threshold=float(db.log_threshold),
             num_keys=db.num_keys, num_entries=db.num_entries, offsets=off.ctypes.data,
             values=db.values.ctypes.data, char_class=cls.ctypes.data, device=0, shard=0)
    d.update(over)
    return capi.PlacerDesc(**d), (off, cls)


@pytest.mark.parametrize("over,code", [
    (dict(abi_version=99), capi.ERR_INVALID),
    (dict(kmer_size=0), capi.ERR_UNSUPPORTED),
    (dict(keep_at_most=0), capi.ERR_UNSUPPORTED),
    (dict(keep_at_most=65), capi.ERR_UNSUPPORTED),
    (dict(offset_bits=16), capi.ERR_INVALID),
    (dict(num_keys=17), capi.ERR_INVALID),
    (dict(num_entries=3), capi.ERR_INVALID),
    (dict(offsets=None), capi.ERR_INVALID),
    # k-mer codes are 32-bit on the device: nucl k <= 15 (4^16 = 2^32 codes is one too many), amino k <= 7
    (dict(kmer_size=16, num_keys=4 ** 16), capi.ERR_UNSUPPORTED),
    (dict(alphabet_size=20, kmer_size=8, num_keys=20 ** 8), capi.ERR_UNSUPPORTED),
])
def test_create_rejects_bad_descriptors(small_case, over, code):
    _, db = small_case
    lib = capi.load()
    desc, keep = _desc(db, **over)
    handle = ctypes.c_void_p()
    rc = lib.epik_amd_placer_create(ctypes.byref(desc), ctypes.byref(handle))
    assert rc == code, lib.epik_amd_last_error()
    assert not handle.value
    assert lib.epik_amd_last_error()


@pytest.mark.parametrize("damage,message", [
    ("duplicate_branch", b"same branch twice"),
    ("branch_out_of_range", b"branch >= num_branches"),
    ("infinite_score", b"non-finite score"),
    ("offsets_not_monotone", b"not monotone"),
])
def test_create_rejects_bad_lists(small_case, damage, message):
    """Host-only checks of the posting lists, made before any device is touched (so they run here)."""
    import numpy as np
    _, db = small_case
    lib = capi.load()
    offsets = np.ascontiguousarray(db.offsets, dtype=np.uint32).copy()
    values = db.values.copy()
    key = int(np.nonzero(np.diff(offsets.astype(np.int64)) >= 2)[0][1])   # a list of two postings or more, not the first
    b = int(offsets[key])
    if damage == "duplicate_branch":
        values["branch"][b + 1] = values["branch"][b]
    elif damage == "branch_out_of_range":
        values["branch"][b] = db.num_branches
    elif damage == "infinite_score":
        values["score"][b] = -np.inf
    else:
        offsets[key + 1] = offsets[key] - 1   # this list would end before it begins
    desc, keep = _desc(db, offsets=offsets.ctypes.data, values=values.ctypes.data)
    handle = ctypes.c_void_p()
    rc = lib.epik_amd_placer_create(ctypes.byref(desc), ctypes.byref(handle))
    assert rc == capi.ERR_INVALID and not handle.value
    assert message in lib.epik_amd_last_error(), lib.epik_amd_last_error()


def test_no_gpu_means_loud_failure_not_cpu_fallback(small_case):
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    _, db = small_case
    lib = capi.load()
    desc, keep = _desc(db)
    handle = ctypes.c_void_p()
    rc = lib.epik_amd_placer_create(ctypes.byref(desc), ctypes.byref(handle))
    assert rc == capi.ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.epik_amd_last_error()
    from epik_amd.placer import Placer
    with pytest.raises(capi.EpikAmdError):
        Placer.from_synth(db)


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under epik_amd/ may reference it."""
    pkg = os.path.join(ROOT, "epik_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "epik_oracle" not in text, f


def test_generated_isa_keeps_its_hands_off_the_load_ring():
    """The posting loads are issued from inline asm and waited for with counted s_waitcnt: hipcc must
    not touch a ring register while its load is in flight.  `make asm` regenerates the ISA of every
    kernel variant and runs epik_amd/csrc/lint_ring_asm.py over it (hipcc cross-compiles here)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    run = subprocess.run(["make", "-C", os.path.join(root, "epik_amd", "csrc"), "asm"], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    # place_kernel.hip, team_kernel.hip, team_stream.hip (more lines when the library had to be rebuilt: its own rule lints too)
    assert run.stdout.count("ring-asm lint: 0 problem(s)") >= 3 and "problem(s)" not in run.stdout.replace("0 problem(s)", "")


def test_the_streaming_kernels_keep_their_register_budgets():
    """The hardware fills a CU by the registers a kernel USES: the lean build of the streaming
    kernel owes its twenty waves per CU to 96 registers and no scratch, the wide build its three workgroups to at most
    168 and none, the front kernel its eight waves per SIMD to 64 (six, 85, when it also sizes the partial lists), the one-wavefront kernel its five to 96.  A source
    or compiler change that breaks one of them costs tens of percent on some tree size without failing anything else."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (the listings the library was linked from: csrc/Makefile compiles with -save-temps and keeps them)
    run = subprocess.run(["make", "-j4", "-C", os.path.join(root, "epik_amd", "csrc")], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    listing = {f: os.path.join(root, "epik_amd", "csrc", "build", "lib", f + "-hip-amdgcn-amd-amdhsa-gfx950.s")
               for f in ("place_kernel", "team_stream")}

    def kernels(path):
        text = open(path).read()
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
            field = lambda k: int(re.search(r"\.amdhsa_" + k + r"\s+(\d+)", m.group(2)).group(1))
            yield m.group(1), field("next_free_vgpr"), field("private_segment_fixed_size")

    seen = 0
    for name, vgpr, scratch in kernels(listing["team_stream"]):
        m = re.match(r"_ZN8epik_amd18team_stream_kernelILi(\d)E([htj])Li(\d)ELb([01])ELi[24]E", name)
        if m and m.group(3) == "0":  # the one-pass placement
            wide, counts = m.group(4) == "1", m.group(2)
            if wide:
                assert vgpr <= 168 and (scratch == 0 or counts == "j"), (name, vgpr, scratch)
            elif counts == "h":
                assert vgpr <= 96 and scratch == 0, (name, vgpr, scratch)
            else:
                assert vgpr <= 128, (name, vgpr, scratch)   # 16- and 32-bit counts: four waves per SIMD at least
            seen += 1
        f = re.match(r"_ZN8epik_amd17team_front_kernelILi(\d)ELb([01])E", name)
        if f and int(f.group(1)) <= 4:  # (eight slices per pass: one workgroup per SIMD asked, what fits runs)
            assert vgpr <= (85 if f.group(2) == "1" else 64), (name, vgpr)   # six / eight waves per SIMD
            seen += 1
    assert seen >= 12
    n = 0
    for name, vgpr, _ in kernels(listing["place_kernel"]):
        if name.startswith("_ZN8epik_amd18place_reads_kernel"):
            assert vgpr <= 96, (name, vgpr)
            n += 1
    assert n >= 6


def _load_lint():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lint_ring_asm", os.path.join(root, "epik_amd", "csrc", "lint_ring_asm.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    return lint


def _issue(score, cell):
    return ["\t;;#ASMSTART", "\ts_nop 4", f"\tbuffer_load_dword v{score}, v1, s[4:7], 0 offen",
            f"\tbuffer_load_ushort v{cell}, v2, s[4:7], s3 offen", "\t;;#ASMEND"]


def _consume(cell, out, count):
    return ["\t;;#ASMSTART", f"\ts_waitcnt vmcnt({count})", f"\tv_mad_i32_i24 v{out}, v{cell}, -4, s1", "\t;;#ASMEND"]


def _ring_kernel(tmp_path, lint, in_first_trip="", in_loop="", loop_wait=0, in_arm=""):
    """A two-slot ring in the shape hipcc emits: a first trip of loads alone, slot 0 waited for, then the loop -- a
    stage per slot: LDS read, wait for the OTHER slot (this one is out and not yet refilled: nothing else in flight,
    vmcnt(0)), LDS write, refill.  Stage 0's write sits in a branch of its own (two paths through the loop)."""
    lines = ["_ZN8epik_amd18place_reads_kernelIcEEvNS_11PlaceParamsE:", "\ts_load_dwordx2 s[0:1], s[4:5], 0x0"]
    lines += _issue(6, 4) + ([in_first_trip] if in_first_trip else []) + _issue(7, 5)   # the first trip: loads alone
    lines += _consume(4, 10, 2) + [".LBB0_1:"]
    lines += ["\tds_read_b32 v12, v10"] + _consume(5, 11, loop_wait)
    lines += ["\ts_cbranch_scc1 .LBB0_2", "\tds_write_b32 v10, v12"] + ([in_arm] if in_arm else []) + [".LBB0_2:"]
    lines += _issue(6, 4) + ([in_loop] if in_loop else [])
    lines += ["\tds_read_b32 v12, v11"] + _consume(4, 10, loop_wait)
    lines += ["\tds_write_b32 v11, v12"] + _issue(7, 5) + ["\ts_cbranch_scc1 .LBB0_1"]
    lines += ["\t;;#ASMSTART", "\ts_waitcnt vmcnt(0)", "\tv_mad_i32_i24 v11, v5, -4, s1", "\t;;#ASMEND", "\tv_mov_b32_e32 v30, v6", "\ts_endpgm"]
    path = tmp_path / "k.s"
    path.write_text("\n".join(lines) + "\n")
    return lint.lint(str(path))


def test_the_ring_lint_sees_a_copy_of_a_register_in_flight(tmp_path):
    """The lint itself (hazard 1, the hardware's rule over the control-flow graph): the ring passes; the same loop with
    one compiler-made copy of a slot register between its load and the wait that retires it is reported -- in the
    loop, in one arm of a branch inside it, and in the first trip of loads alone that fills an empty ring."""
    lint = _load_lint()
    assert _ring_kernel(tmp_path, lint) == []
    assert len(_ring_kernel(tmp_path, lint, in_loop="\tv_mov_b32_e32 v20, v4")) == 1         # slot 0 was refilled a moment ago
    assert len(_ring_kernel(tmp_path, lint, in_first_trip="\tv_mov_b32_e32 v20, v6")) == 1    # slot 0's score: its load has just been issued
    assert _ring_kernel(tmp_path, lint, in_first_trip="\tv_mov_b32_e32 v20, v7") == []        # slot 1's registers are still free there
    # (one arm of the branch inside the loop: slot 0 is out there and not yet refilled -- free; slot 1 has been waited for)
    assert _ring_kernel(tmp_path, lint, in_arm="\tv_mov_b32_e32 v20, v4") == []
    assert _ring_kernel(tmp_path, lint, in_arm="\tv_mov_b32_e32 v20, v5") == []


def test_the_ring_lint_checks_the_counted_waits_themselves(tmp_path):
    """A wait that counts one load too many leaves the slot it was meant for in flight: the asm instruction behind it
    that reads the slot is reported.  So is the same correct count when a load the ring did not issue -- a
    compiler-made one -- has slipped in behind the slot's: vmcnt(N) then retires one load fewer of the ring's."""
    lint = _load_lint()
    too_long = _ring_kernel(tmp_path, lint, loop_wait=2)     # vmcnt(2) with two loads in flight waits for nothing
    assert too_long and all("v_mad_i32_i24" in p for p in too_long)
    # a compiler load between slot 0's refill and the next wait: the wait's count no longer covers slot 1 ... it does
    # (vmcnt(0) drains everything); with counted waits that leave loads in flight it would not:
    lines = ["_ZN8epik_amd18place_reads_kernelIcEEvNS_11PlaceParamsE:"]
    lines += _issue(6, 4) + _issue(7, 5) + _consume(4, 10, 2)                 # slot 0 is retired: two loads behind its cell
    ok = lines + ["\ts_endpgm"]
    slipped = lines[:1] + _issue(6, 4) + ["\tglobal_load_dword v40, v[2:3], off"] + _issue(7, 5)
    slipped += ["\t;;#ASMSTART", "\ts_waitcnt vmcnt(3)", "\tv_mad_i32_i24 v10, v4, -4, s1", "\t;;#ASMEND", "\ts_endpgm"]   # right again: three behind
    short = lines[:1] + _issue(6, 4) + _issue(7, 5) + ["\tglobal_load_dword v40, v[2:3], off"]
    short += ["\t;;#ASMSTART", "\ts_waitcnt vmcnt(3)", "\tv_mad_i32_i24 v10, v4, -4, s1", "\t;;#ASMEND", "\ts_endpgm"]   # three behind: retired
    stale = lines[:1] + _issue(6, 4) + _issue(7, 5)
    stale += ["\t;;#ASMSTART", "\ts_waitcnt vmcnt(3)", "\tv_mad_i32_i24 v10, v4, -4, s1", "\t;;#ASMEND", "\ts_endpgm"]   # only two behind: in flight

    def run(body):
        path = tmp_path / "q.s"
        path.write_text("\n".join(body) + "\n")
        return lint.lint(str(path))
    assert run(ok) == [] and run(slipped) == [] and run(short) == []
    assert len(run(stale)) == 1


def test_the_ring_lint_counts_the_wait_states_in_front_of_a_refill(tmp_path):
    """A refill without its s_nop 4 (the stages of the ring: Layout::issue<kSettled>) must lie five wait states
    behind the v_readlane that made its buffer resource -- in straight-line code, through the local labels of the
    run-coded refill's two arms, and through a basic-block label of the compiler's (over every path that reaches it)."""
    lint = _load_lint()

    def problems(between, load):
        lines = ["_ZN8epik_amd18place_reads_kernelIcEEvNS_11PlaceParamsE:", "\ts_load_dwordx2 s[0:1], s[4:5], 0x0",
                 "\tv_readlane_b32 s9, v2, 0", "\tv_readlane_b32 s4, v3, 0"] + between + load + ["\ts_endpgm"]
        out = []
        lint.lint_settled("k", lines, out)
        return out

    filler = ["\ts_lshr_b32 s10, s9, 16", "\ts_and_b32 s5, s9, 0xffff", "\ts_mul_i32 s6, s10, 6", "\ts_lshl_b32 s3, s10, 2"]
    plain = ["\tbuffer_load_dword v6, v1, s[4:7], 0 offen", "\tbuffer_load_ushort v4, v2, s[4:7], s3 offen"]
    assert len(problems(filler, plain)) == 1                                  # four instructions between: one short
    assert problems(filler + ["\tv_add_f32 v1, v2, v3"], plain) == []
    assert problems(filler[:1] + ["\ts_nop 3"], plain) == []                   # s_nop 3: four wait states
    assert len(problems(filler[:1] + ["\ts_nop 2"], plain)) == 1
    # the two arms of the run-coded refill: the arm behind the label is reached from the branch
    arms = ["\ts_cmp_eq_u32 s8, 0", "\ts_cbranch_scc1 .Lexplicit7", "\tbuffer_load_dword v6, v1, s[4:7], 0 offen",
            "\tv_sub_u32 v4, s8, v0", "\ts_branch .Lissued7", ".Lexplicit7:", "\tbuffer_load_ushort v4, v2, s[4:7], s3 offen",
            "\tbuffer_load_dword v6, v1, s[4:7], 0 offen", ".Lissued7:"]
    assert len(problems(filler[:2], arms)) == 2      # both arms four wait states behind the v_readlane
    assert problems(filler[:3], arms) == []
    # a basic-block label of the compiler's between the v_readlane and the load: the load is checked against what
    # precedes the label on every path -- the fall-through and the branch that jumps to it
    assert len(problems(filler[:2] + [".LBB0_9:"], plain)) == 2     # (both loads of the chunk)
    assert problems(filler + [".LBB0_9:", "\ts_nop 0"], plain) == []
    jump = ["\ts_cbranch_scc1 .LBB0_9"] + filler + ["\ts_nop 0", ".LBB0_9:"]   # the fall-through is long enough, the jump is not
    assert len(problems(jump, plain)) == 2


def test_release_scratch_rejects_null():
    lib = capi.load()
    assert lib.epik_amd_placer_release_scratch(None) == capi.ERR_INVALID


def test_two_hip_runtimes_are_named_not_left_to_fail_later():
    """libepik_amd.so links the system's libamdhip64 by SONAME, torch ships its own: loaded in the wrong order the
    process holds two runtimes and torch says "No HIP GPUs" much later.  capi.load() / device_count() say so."""
    import subprocess
    import sys
    from epik_amd import capi
    maps = ("7f00-7f01 r-xp 0 00:00 1 /opt/rocm-7.2.0/lib/libamdhip64.so.7.2.70200\n"
            "7f02-7f03 r--p 0 00:00 2 /opt/rocm-7.2.0/lib/libamdhip64.so.7.2.70200\n"
            "7f04-7f05 r-xp 0 00:00 3 /usr/lib/python3/dist-packages/torch/lib/libamdhip64.so\n"
            "7f06-7f07 r-xp 0 00:00 4 /usr/lib/libc.so.6\n")
    found = capi.hip_runtimes(maps)
    assert len(found) == 2 and all("libamdhip64" in f for f in found)
    with pytest.raises(ImportError, match="Import torch BEFORE"):
        capi.check_hip_runtime(found)
    capi.check_hip_runtime(found[:1])
    capi.check_hip_runtime([])
    # the real thing, in a process of its own: our library first, torch second
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from epik_amd import capi\n"
            "capi.load()\n"
            "import torch\n"
            "try:\n"
            "    capi.device_count()\n"
            "except ImportError as err:\n"
            "    print('REFUSED', err)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    if len(capi.hip_runtimes()) <= 1 and "REFUSED" not in out.stdout:
        # (a torch built against the system's ROCm brings no runtime of its own: nothing to refuse)
        assert out.returncode == 0, out.stderr[-2000:]
    else:
        assert "REFUSED" in out.stdout and "two HIP runtimes" in out.stdout, out.stdout + out.stderr[-2000:]


def test_a_library_without_a_passed_lint_is_refused(tmp_path):
    """capi.load() and smoke() refuse a library whose build record does not carry the ISA lint's mark (built by hand, or
    by something other than the Makefile's lint-then-link rule), and smoke() also one older than the sources."""
    import json
    import subprocess
    import sys
    from epik_amd import provenance
    record = os.path.join(ROOT, "epik_amd", "libepik_amd.build.json")
    good = open(record).read()
    assert provenance.summary()["lint_covers_this_build"]
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from epik_amd import capi\n"
            "capi.load()\nprint('LOADED')\n" % ROOT)
    try:
        doc = json.loads(good)
        doc.pop("lint")
        with open(record, "w") as fh:
            json.dump(doc, fh)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "no record of a passed ISA lint" in out.stderr and "LOADED" not in out.stdout
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, EPIK_AMD_ALLOW_UNLINTED="1"))
        assert "LOADED" in out.stdout
        doc = json.loads(good)
        doc["kernel_source_hash"] = "0" * 16   # a library older than the tree's sources
        with open(record, "w") as fh:
            json.dump(doc, fh)
        with pytest.raises(ImportError, match="rebuild it"):
            provenance.check_library(strict_sources=True)
        provenance.check_library()   # (the plain load does not mind: the CPU suite runs while sources are being edited)
    finally:
        with open(record, "w") as fh:
            fh.write(good)


def test_build_record_names_the_compiler_and_the_sources():
    from epik_amd import provenance
    summary = provenance.summary()
    assert summary["kernel_source_hash"] == provenance.kernel_source_hash()
    built = provenance.built_with()
    assert built, "make -C epik_amd/csrc writes epik_amd/libepik_amd.build.json"
    assert "clang" in built["hipcc"] or "HIP version" in built["hipcc"], built
    assert summary["library_is_current"], "the library in the tree was not built from the sources in the tree"
    lint = provenance.lint_record()
    assert lint and lint["hipcc"] == built["hipcc"], "csrc/lint_passed.json: `make -C epik_amd/csrc asm` with this compiler"
