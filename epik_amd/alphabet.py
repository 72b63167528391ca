"""Character-class tables for the k-mer encoder (the i2l `seq_traits` side).

The reference encodes k-mers inside the absent `i2l` submodule
(`i2l::to_kmers<one_ambiguity_policy>`, call site place.cpp:294), so the state
order below is an ASSUMPTION recalled from upstream i2l, not verifiable in this
image (SURVEY.md 2.2, 8c "assumption register").  It is isolated here: the device
kernel, the C-ABI and the oracle only ever see the 256-entry table
`char_class[c]` = bitmask of the states character `c` may stand for
(popcount 1 = plain state, >1 = ambiguous, 0 = invalid), so swapping the order
means editing this file only.
"""
from __future__ import annotations

import ctypes
import ctypes.util

import numpy as np

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.log10f.restype = ctypes.c_float
_libm.log10f.argtypes = [ctypes.c_float]

#: nucleotide state order: A=0 C=1 G=2 T=3 (2 bits per character, first character
#: most significant); U is T; case-insensitive.
NUCL_STATES = "ACGT"
NUCL_AMBIGUOUS = {
    "R": "AG", "Y": "CT", "S": "CG", "W": "AT", "K": "GT", "M": "AC",
    "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG", "N": "ACGT",
}

#: amino-acid state order recalled from i2l's `seq_traits<aa>` (unverified).
AMINO_STATES = "RHKDESTNQCGPAILMFWYV"
AMINO_AMBIGUOUS = {"B": "DN", "Z": "EQ", "J": "IL", "X": AMINO_STATES}


def _table(states: str, ambiguous: dict[str, str], aliases: dict[str, str]) -> np.ndarray:
    table = np.zeros(256, dtype=np.uint32)
    index = {c: i for i, c in enumerate(states)}
    for c, i in index.items():
        table[ord(c)] = 1 << i
        table[ord(c.lower())] = 1 << i
    for c, members in ambiguous.items():
        mask = 0
        for m in members:
            mask |= 1 << index[m]
        table[ord(c)] = mask
        table[ord(c.lower())] = mask
    for c, target in aliases.items():
        table[ord(c)] = table[ord(target)]
        table[ord(c.lower())] = table[ord(target)]
    return table


def char_class_table(states: str) -> np.ndarray:
    """Returns the uint32[256] class table for `states` in {'nucl', 'amino'}
    (the `-s/--states` choice of epik.py:33-37)."""
    if states == "nucl":
        return _table(NUCL_STATES, NUCL_AMBIGUOUS, {"U": "T"})
    if states == "amino":
        return _table(AMINO_STATES, AMINO_AMBIGUOUS, {})
    raise ValueError(f"unknown states {states!r}: expected 'nucl' or 'amino'")


def alphabet_size(states: str) -> int:
    return {"nucl": 4, "amino": 20}[states]


def state_chars(states: str) -> str:
    return {"nucl": NUCL_STATES, "amino": AMINO_STATES}[states]


def score_threshold(omega: float, kmer_size: int, sigma: int) -> np.float32:
    """`i2l::score_threshold(omega, k)` (call site place.cpp:87): a probability,
    (omega / sigma)^k as float32.  ASSUMPTION (i2l absent): evaluated in double
    and rounded once."""
    return np.float32((float(np.float32(omega)) / float(sigma)) ** int(kmer_size))


def log_threshold(threshold: np.float32) -> np.float32:
    """`std::log10(_threshold)` on a float (place.cpp:88) = glibc log10f (called
    through libm so that Python and the C++ host agree bit for bit)."""
    return np.float32(_libm.log10f(ctypes.c_float(float(threshold))))
