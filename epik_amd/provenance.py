"""Ties a measurement to the kernel sources it was taken on: profiles/traffic.json carries the hash of
the files below as they were during its PMC passes, and bench.py reports that traffic only while the
hash still matches (a kernel change that alters the traffic must not keep reporting the old number)."""
from __future__ import annotations

import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (the kernels the counter passes measure, the image builder that lays out what they read, and the launch code that
# sizes their grids; shard_place.hip -- the host side of place_sharded and its one element-wise kernel -- is not among
# them: no counter pass runs it)
KERNEL_SOURCES = ("place_kernel.hip", "place_device.hpp", "team_kernel.hip", "team_stream.hip", "team_device.hpp",
                  "team_epilogue.hpp", "place_kernel.h", "db_layout.h", "db_image.cpp", "capi.hip", "placer_impl.hpp")


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "epik_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def built_with() -> dict:
    """What `make -C epik_amd/csrc` recorded next to the library it built (epik_amd/libepik_amd.build.json): the
    compiler (`hipcc --version`: the hot loop's registers are invisible to hipcc and csrc/lint_ring_asm.py checks the
    ISA a given compiler made of it), the hash of the kernel sources at that time and whether the lint passed.
    {} when the record is missing (a library built by hand)."""
    import json
    try:
        with open(os.path.join(ROOT, "epik_amd", "libepik_amd.build.json")) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


def lint_record() -> dict:
    """epik_amd/csrc/lint_passed.json: the compiler and sources csrc/lint_ring_asm.py last accepted the ISA of."""
    import json
    try:
        with open(os.path.join(ROOT, "epik_amd", "csrc", "lint_passed.json")) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


def summary() -> dict:
    """For bench.py's line, smoke() and capi.load(): the compiler the library was built with, the sources it was built
    from against the sources now, and whether the ISA lint passed on exactly that build -- the Makefile links the
    library only from kernel objects whose ISA (hipcc -save-temps: the very listing the object was assembled from)
    lint_ring_asm.py accepted, and says so in the record it writes next to the library.  `lint_passed_on`: the
    committed record of the last `make asm` (history: which compiler and sources the lint was last seen green on)."""
    built, lint, now = built_with(), lint_record(), kernel_source_hash()
    return {"hipcc": built.get("hipcc"), "built_from": built.get("kernel_source_hash"), "kernel_source_hash": now,
            "library_is_current": built.get("kernel_source_hash") == now,
            "lint_passed_on": {"hipcc": lint.get("hipcc"), "kernel_source_hash": lint.get("kernel_source_hash")},
            "lint_covers_this_build": bool(built.get("lint")) and bool(built.get("hipcc"))}


def check_library(strict_sources: bool = False) -> dict:
    """Refuses a library the ISA lint did not cover (no build record next to it, or one without the lint's mark: built by
    hand, or by a Makefile older than the rule): its streaming loop may read registers whose loads are in flight, and
    nothing else would notice.  EPIK_AMD_ALLOW_UNLINTED=1 lets it through (experiments: tools/ablate.py variants).
    `strict_sources`: also refuse a library older than the kernel sources beside it (smoke(), the GPU tests: a stale
    binary would pass or fail for the wrong sources)."""
    s = summary()
    if not s["lint_covers_this_build"] and os.environ.get("EPIK_AMD_ALLOW_UNLINTED") != "1":
        raise ImportError(
            "epik_amd/libepik_amd.so has no record of a passed ISA lint (epik_amd/libepik_amd.build.json: "
            f"{built_with() or 'missing'}).  Build it with `make -C epik_amd/csrc` -- the Makefile lints the ISA hipcc made of "
            "the streaming kernels before it links (csrc/lint_ring_asm.py) -- or set EPIK_AMD_ALLOW_UNLINTED=1.")
    if strict_sources and not s["library_is_current"]:
        raise ImportError(
            f"epik_amd/libepik_amd.so was built from kernel sources {s['built_from']}, the tree holds {s['kernel_source_hash']}: "
            "rebuild it (`make -C epik_amd/csrc`).")
    return s
