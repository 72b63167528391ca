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
                  "place_kernel.h", "db_layout.h", "db_image.cpp", "capi.hip", "placer_impl.hpp")


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "epik_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]
