"""ctypes binding of libepik_amd.so -- the C-ABI declared in include/epik_amd.h.

This is the same stub a maintainer of another host language would write (see
INTEGRATION.md).  The library is built in-tree by `__graft_entry__.build()` /
`make -C epik_amd/csrc`.  Loading fails loudly when it is missing: there is no
Python or CPU fallback for the placement path.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EPIK_AMD_LIB") or os.path.join(_HERE, "libepik_amd.so")

ABI_VERSION = 3

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED = 0, 1, 2, 3, 4

#: numpy mirror of `epik_amd_placement` {branch, score, lwr} (16 bytes)
PLACEMENT = np.dtype([("branch", np.uint32), ("score", np.float32), ("lwr", np.float64)])
#: numpy mirror of `epik_amd_pkdb_value` / `i2l::pkdb_value` (8 bytes)
PKDB_VALUE = np.dtype([("branch", np.uint32), ("score", np.float32)])

#: every symbol include/epik_amd.h declares
EXPORTS = (
    "epik_amd_device_count",
    "epik_amd_last_error",
    "epik_amd_placer_create",
    "epik_amd_placer_plan",
    "epik_amd_placer_plan_sizes",
    "epik_amd_placer_build_image",
    "epik_amd_placer_destroy",
    "epik_amd_placer_place",
    "epik_amd_placer_place_device",
    "epik_amd_placer_algorithmic_bytes",
    "epik_amd_placer_create_sharded",
    "epik_amd_placer_accumulate_device",
    "epik_amd_placer_finish_device",
    "epik_amd_placer_partial_info",
    "epik_amd_placer_accumulate_lists_device",
    "epik_amd_placer_finish_lists_device",
    "epik_amd_placer_last_path",
    "epik_amd_placer_stream_build",
    "epik_amd_placer_place_sharded",
    "epik_amd_placer_release_scratch",
    "epik_amd_placer_set_wide_counts",
    "epik_amd_placer_choose_counts",
    "epik_amd_placer_launch_info",
    "epik_amd_placer_set_timing",
    "epik_amd_placer_last_kernel_ms",
)


class PlacerDesc(ctypes.Structure):
    """`epik_amd_placer_desc`."""

    _fields_ = [
        ("abi_version", ctypes.c_uint32),
        ("kmer_size", ctypes.c_uint32),
        ("alphabet_size", ctypes.c_uint32),
        ("num_branches", ctypes.c_uint32),
        ("keep_at_most", ctypes.c_uint32),
        ("offset_bits", ctypes.c_uint32),
        ("keep_factor", ctypes.c_double),
        ("threshold", ctypes.c_float),
        ("log_threshold", ctypes.c_float),
        ("num_keys", ctypes.c_uint64),
        ("num_entries", ctypes.c_uint64),
        ("offsets", ctypes.c_void_p),
        ("values", ctypes.c_void_p),
        ("char_class", ctypes.c_void_p),
        ("device", ctypes.c_int32),
        ("shard", ctypes.c_uint32),
        ("keys", ctypes.c_void_p),
        ("num_present", ctypes.c_uint64),
    ]


class Plan(ctypes.Structure):
    """`epik_amd_plan`."""

    _fields_ = [
        ("kernel", ctypes.c_uint32),
        ("layout", ctypes.c_uint32),
        ("team_waves", ctypes.c_uint32),
        ("team_passes", ctypes.c_uint32),
        ("slice_rows", ctypes.c_uint32),
        ("resident_waves", ctypes.c_uint32 * 3),
        ("table_bytes", ctypes.c_uint64),
        ("filter_bytes", ctypes.c_uint64),
        ("posting_bytes", ctypes.c_uint64),
        ("kept_entries", ctypes.c_uint64),
        ("run_coded", ctypes.c_uint32),
        ("posting_bytes_is_bound", ctypes.c_uint32),
    ]


class ListBin(ctypes.Structure):
    """`epik_amd_list_bin`."""

    _fields_ = [("length", ctypes.c_uint64), ("lists", ctypes.c_uint64), ("lists_in_runs", ctypes.c_uint64)]


class PartialInfo(ctypes.Structure):
    """`epik_amd_partial_info`."""

    _fields_ = [
        ("lists", ctypes.c_uint32),
        ("slices", ctypes.c_uint32),
        ("slice_rows", ctypes.c_uint32),
        ("entry_bytes", ctypes.c_uint32),
        ("num_branches", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
        ("postings_per_kmer", ctypes.c_double),
    ]


MAX_SHARDS = 16
#: count of a partial list that found no room in d_entries
LIST_OVERFLOW = 0xFFFFFFFF
PATH_WAVE, PATH_TEAM_ONE_KERNEL, PATH_TEAM_STREAMED = 0, 1, 2

#: n_rows of a read with more k-mers than the counts of a device-pointer launch hold
ROWS_COUNTS_TOO_NARROW = 0xFFFFFFFF


class EpikAmdError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libepik_amd error {code}: {message}")
        self.code = code


_lib = None


def hip_runtimes(maps_text: str | None = None) -> list:
    """The HIP runtimes (libamdhip64) mapped into this process, from /proc/self/maps (or `maps_text`)."""
    import re
    if maps_text is None:
        try:
            with open("/proc/self/maps") as fh:
                maps_text = fh.read()
        except OSError:
            return []
    found = {os.path.realpath(m.group(1)) for m in re.finditer(r"(/\S*libamdhip64\S*)", maps_text)}
    return sorted(found)


def check_hip_runtime(runtimes: list | None = None) -> None:
    """libepik_amd.so links the system's HIP runtime by SONAME; PyTorch-ROCm ships one of its own.  When torch is
    imported FIRST its runtime satisfies the SONAME and the process holds one runtime (what bench.py, the tests and
    every caller that hands torch tensors to this library do).  The other way round the process ends up with TWO
    runtimes -- two device contexts, and torch reports "No HIP GPUs" at some later point.  Said here, by name."""
    runtimes = hip_runtimes() if runtimes is None else runtimes
    if len(runtimes) > 1:
        raise ImportError(
            "two HIP runtimes are mapped into this process (" + ", ".join(runtimes) + "): libepik_amd.so was loaded "
            "before torch, whose own libamdhip64 then came on top of the system's.  Import torch BEFORE epik_amd.capi.load() "
            "(`import torch` at the top of the program), or use a torch built against the system's ROCm.")


def load() -> ctypes.CDLL:
    """Loads libepik_amd.so (once) and declares the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C epik_amd/csrc` (hipcc, gfx950).  epik_amd has no CPU fallback.")
    if not os.environ.get("EPIK_AMD_LIB"):  # (a library named by hand is the caller's: tools/ablate.py variants)
        from . import provenance
        provenance.check_library()  # no record of a passed ISA lint: refused
    lib = ctypes.CDLL(LIB_PATH)
    check_hip_runtime()
    vp, u64, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int
    lib.epik_amd_device_count.restype = i32
    lib.epik_amd_device_count.argtypes = []
    lib.epik_amd_last_error.restype = ctypes.c_char_p
    lib.epik_amd_last_error.argtypes = []
    lib.epik_amd_placer_create.restype = i32
    lib.epik_amd_placer_create.argtypes = [ctypes.POINTER(PlacerDesc), ctypes.POINTER(vp)]
    lib.epik_amd_placer_create_sharded.restype = i32
    lib.epik_amd_placer_create_sharded.argtypes = [ctypes.POINTER(PlacerDesc), ctypes.c_uint32, ctypes.c_uint32,
                                                   ctypes.POINTER(vp)]
    lib.epik_amd_placer_accumulate_device.restype = i32
    lib.epik_amd_placer_accumulate_device.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp, vp, vp]
    lib.epik_amd_placer_finish_device.restype = i32
    lib.epik_amd_placer_finish_device.argtypes = [vp, vp, u64, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.epik_amd_placer_partial_info.restype = i32
    lib.epik_amd_placer_partial_info.argtypes = [vp, ctypes.POINTER(PartialInfo)]
    lib.epik_amd_placer_accumulate_lists_device.restype = i32
    lib.epik_amd_placer_accumulate_lists_device.argtypes = [vp, vp, vp, u64, ctypes.c_uint32, vp, u64, vp, vp, vp, vp, vp, vp]
    lib.epik_amd_placer_finish_lists_device.restype = i32
    lib.epik_amd_placer_finish_lists_device.argtypes = [vp, vp, u64, ctypes.c_uint32, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                                        vp, vp, vp, vp, vp, vp]
    lib.epik_amd_placer_place_sharded.restype = i32
    lib.epik_amd_placer_place_sharded.argtypes = [ctypes.POINTER(vp), ctypes.c_uint32, vp, vp, u64, vp, vp, vp]
    lib.epik_amd_placer_release_scratch.restype = i32
    lib.epik_amd_placer_release_scratch.argtypes = [vp]
    lib.epik_amd_placer_stream_build.restype = i32
    lib.epik_amd_placer_stream_build.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    lib.epik_amd_placer_last_path.restype = i32
    lib.epik_amd_placer_last_path.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32)]
    lib.epik_amd_placer_plan.restype = i32
    lib.epik_amd_placer_plan.argtypes = [ctypes.POINTER(PlacerDesc), ctypes.c_uint32, ctypes.c_uint32, u64,
                                         ctypes.POINTER(Plan)]
    lib.epik_amd_placer_plan_sizes.restype = i32
    lib.epik_amd_placer_plan_sizes.argtypes = [ctypes.c_uint32] * 4 + [ctypes.POINTER(ListBin), u64, ctypes.c_uint32, ctypes.c_uint32,
                                               u64, ctypes.POINTER(Plan)]
    lib.epik_amd_placer_build_image.restype = i32
    lib.epik_amd_placer_build_image.argtypes = [ctypes.POINTER(PlacerDesc), ctypes.c_uint32, ctypes.c_uint32, u64,
                                                vp, vp, vp]
    lib.epik_amd_placer_destroy.restype = None
    lib.epik_amd_placer_destroy.argtypes = [vp]
    lib.epik_amd_placer_place.restype = i32
    lib.epik_amd_placer_place.argtypes = [vp, vp, vp, u64, vp, vp, vp]
    lib.epik_amd_placer_place_device.restype = i32
    lib.epik_amd_placer_place_device.argtypes = [vp, vp, vp, u64, vp, vp, vp, vp]
    lib.epik_amd_placer_algorithmic_bytes.restype = i32
    lib.epik_amd_placer_algorithmic_bytes.argtypes = [vp, vp, vp, u64, vp, vp, ctypes.POINTER(u64)]
    lib.epik_amd_placer_launch_info.restype = i32
    lib.epik_amd_placer_launch_info.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32),
                                                ctypes.POINTER(ctypes.c_uint32),
                                                ctypes.POINTER(ctypes.c_uint32)]
    lib.epik_amd_placer_set_wide_counts.restype = i32
    lib.epik_amd_placer_set_wide_counts.argtypes = [vp, i32]
    lib.epik_amd_placer_choose_counts.restype = i32
    lib.epik_amd_placer_choose_counts.argtypes = [vp, u64]
    lib.epik_amd_placer_set_timing.restype = i32
    lib.epik_amd_placer_set_timing.argtypes = [vp, i32]
    lib.epik_amd_placer_last_kernel_ms.restype = i32
    lib.epik_amd_placer_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != OK:
        raise EpikAmdError(code, (load().epik_amd_last_error() or b"").decode(errors="replace"))


def device_count() -> int:
    lib = load()
    check_hip_runtime()  # (torch may have been imported since load())
    return int(lib.epik_amd_device_count())
