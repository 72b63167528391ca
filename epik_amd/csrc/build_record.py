#!/usr/bin/env python3
"""Records what built the library and what the ISA lint passed on (called by the Makefile).

    build_record.py build [suffix] -> epik_amd/libepik_amd<suffix>.build.json (travels with the .so, git-ignored):
                               hipcc --version, the hash of the kernel sources the library was built from, and that
                               lint_ring_asm.py passed on the ISA it was linked from (the Makefile links nothing else)
    build_record.py lint    -> epik_amd/csrc/lint_passed.json (committed): the same two facts for the last run of
                               `make asm` in which lint_ring_asm.py accepted the generated ISA of every kernel

The streaming loop's registers are invisible to hipcc (place_device.hpp: stream_round); lint_ring_asm.py is what
stands between a compiler upgrade and silent corruption, so the compiler it last passed on is pinned next to the
sources, and bench.py prints both records and whether they agree."""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def hipcc_version() -> str:
    try:
        out = subprocess.run([os.environ.get("HIPCC", "hipcc"), "--version"], capture_output=True, text=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError) as err:
        return f"unknown ({err})"
    keep = [line.strip() for line in out.splitlines() if line.startswith(("HIP version", "AMD clang version", "Target"))]
    return "; ".join(keep) or out.strip()[:200]


def main():
    from epik_amd import provenance
    what = sys.argv[1] if len(sys.argv) > 1 else "build"
    record = {"hipcc": hipcc_version(), "kernel_source_hash": provenance.kernel_source_hash(),
              "when": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}
    if what == "lint":
        record["lint"] = "lint_ring_asm.py passed on place_kernel.hip, team_kernel.hip, team_stream.hip"
        path = os.path.join(HERE, "lint_passed.json")
        # (the committed record changes when the compiler or the sources do, not with every run of the test suite)
        try:
            with open(path) as fh:
                last = json.load(fh)
            if all(last.get(k) == record[k] for k in ("hipcc", "kernel_source_hash", "lint")):
                return
        except (OSError, ValueError):
            pass
    else:
        suffix = sys.argv[2] if len(sys.argv) > 2 else ""
        record["lint"] = "lint_ring_asm.py passed on the ISA this library was linked from"
        path = os.path.join(ROOT, "epik_amd", f"libepik_amd{suffix}.build.json")
    with open(path, "w") as fh:
        json.dump(record, fh, indent=1)
        fh.write("\n")


if __name__ == "__main__":
    main()
