"""EPIKAMD1 -- this repository's flat container for a phylo-k-mer database.

EPIK's own `.ipk` files are Boost.Serialization archives produced inside the absent
`i2l` library; neither the library nor a sample file is available, so that format can
be neither restated nor validated here.  This container keeps what the
loader's call site needs (`i2l::load(file, mu, omega, max_entries)`, main.cpp:277):
k-mer records stored most informative first (a prefix = the best `mu` fraction), the
tree, k, omega, the sequence type.  The C++ reader is epik_amd/host/phylo_kmer_db.cpp.

Layout (little endian):
    char[8] "EPIKAMD1" | u32 version | u32 sequence_type (0 DNA, 1 Proteins) | u32 k | f32 omega
    u64 num_kmers | u64 num_entries_total | u64 newick_len | char newick[newick_len]
    num_kmers x { u32 key (dense k-mer code) | u32 n | n x { u32 branch, f32 score } }
    version >= 2, the last 32 bytes of the file -- a converter that died half way, or a copy cut short, is caught at
    load, not at the first wrong placement:
    char[8] "EPIKEND1" | u64 num_kmers | u64 num_entries | u32 crc32 of the record bytes (zlib's polynomial) | u32 0
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

from . import alphabet
from .synth import PKDB_VALUE, SynthDB

MAGIC = b"EPIKAMD1"
END_MAGIC = b"EPIKEND1"
VERSION = 2
TRAILER_BYTES = 32


def informativeness_order(offsets: np.ndarray, values: np.ndarray) -> np.ndarray:
    """Order of the present keys, most informative first.  ASSUMPTION: IPK ranks a k-mer by
    how far its branch scores spread; here: best (largest) score of the list, descending,
    ties by key -- any fixed total order gives `mu` / `--max-ram` their prefix semantics."""
    lens = np.diff(offsets.astype(np.int64))
    keys = np.nonzero(lens)[0]
    starts = offsets[keys].astype(np.int64)
    best = np.maximum.reduceat(values["score"], starts) if len(keys) else np.zeros(0, np.float32)
    return keys[np.lexsort((keys, -best.astype(np.float64)))]


def write_db(path: str, db: SynthDB, newick: str) -> None:
    order = informativeness_order(db.offsets, db.values)
    with open(path, "wb") as fh:
        fh.write(MAGIC)
        fh.write(struct.pack("<IIIf", VERSION, 0 if db.states == "nucl" else 1, db.kmer_size,
                             float(db.omega)))
        tree = newick.encode()
        fh.write(struct.pack("<QQQ", len(order), db.num_entries, len(tree)))
        fh.write(tree)
        offs = db.offsets.astype(np.int64)
        crc = 0
        for key in order:
            b, e = int(offs[key]), int(offs[key + 1])
            record = struct.pack("<II", int(key), e - b) + db.values[b:e].tobytes()
            crc = zlib.crc32(record, crc)
            fh.write(record)
        fh.write(END_MAGIC + struct.pack("<QQII", len(order), db.num_entries, crc & 0xFFFFFFFF, 0))


def check_trailer(fh, path: str, num_kmers: int, total: int) -> None:
    """The trailer of a version-2 file against its header and its record bytes (the file position is kept)."""
    at = fh.tell()
    fh.seek(0, 2)
    size = fh.tell()
    if size < at + TRAILER_BYTES:
        raise RuntimeError(f"{path} is truncated: no room for the EPIKEND1 trailer")
    fh.seek(size - TRAILER_BYTES)
    tail = fh.read(TRAILER_BYTES)
    if tail[:8] != END_MAGIC:
        raise RuntimeError(f"{path} is truncated or was not finished: the EPIKEND1 trailer is missing")
    kmers_written, entries_written, crc, _ = struct.unpack("<QQII", tail[8:])
    if kmers_written != num_kmers or entries_written != total:
        raise RuntimeError(f"{path}: the trailer counts {kmers_written} k-mers / {entries_written} phylo-k-mers, the header "
                           f"{num_kmers} / {total}")
    fh.seek(at)
    got, left = 0, size - TRAILER_BYTES - at
    while left > 0:
        block = fh.read(min(left, 1 << 24))
        got = zlib.crc32(block, got)
        left -= len(block)
    if (got & 0xFFFFFFFF) != crc:
        raise RuntimeError(f"{path}: the records do not match their checksum (crc32 {got & 0xFFFFFFFF:08x}, trailer {crc:08x})")
    fh.seek(at)


def read_db(path: str, mu: float = 1.0, omega: float = 1.5, max_entries: int | None = None):
    """Python twin of epik_amd::load (same filtering rules).  Returns (SynthDB, newick)."""
    with open(path, "rb") as fh:
        if fh.read(8) != MAGIC:
            raise RuntimeError(f"{path} is not an EPIKAMD1 file (IPK .ipk files cannot be read by this build)")
        version, seq_type, k, built_omega = struct.unpack("<IIIf", fh.read(16))
        num_kmers, total, newick_len = struct.unpack("<QQQ", fh.read(24))
        newick = fh.read(newick_len).decode()
        if version >= 2:
            check_trailer(fh, path, num_kmers, total)
        states = "nucl" if seq_type == 0 else "amino"
        sigma = alphabet.alphabet_size(states)
        eff_omega = max(float(np.float32(omega)), float(np.float32(built_omega)))
        log_thr = alphabet.log_threshold(alphabet.score_threshold(eff_omega, k, sigma))
        lists = {}
        loaded = 0
        to_load = int(np.ceil(float(np.float32(mu)) * num_kmers))
        for _ in range(min(num_kmers, to_load)):
            key, n = struct.unpack("<II", fh.read(8))
            vals = np.frombuffer(fh.read(8 * n), dtype=PKDB_VALUE)
            vals = vals[vals["score"] >= log_thr]
            if max_entries is not None and loaded + len(vals) > max_entries:
                break
            if len(vals):
                lists[key] = vals
                loaded += len(vals)
    num_keys = sigma ** k
    lens = np.zeros(num_keys, dtype=np.int64)
    for key, vals in lists.items():
        lens[key] = len(vals)
    offsets = np.zeros(num_keys + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:].view(np.int64))
    values = np.zeros(int(offsets[-1]), dtype=PKDB_VALUE)
    for key, vals in lists.items():
        values[int(offsets[key]):int(offsets[key + 1])] = vals
    n_branches = newick.count(",") * 2 + 1 if newick else 0
    db = SynthDB(states=states, kmer_size=k, omega=eff_omega, num_branches=n_branches,
                 offsets=offsets, values=values)
    return db, newick
